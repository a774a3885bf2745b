// Domain-decomposition planner behind the C-ABI (host-only, integer work): which nodes of a rank's local mesh it owns, which it
// needs as ghosts, the [owned | ghost] renumbering and the send lists of the ghost exchange -- what the reference derives in
// Mesh::dofmap_* (lowest rank touching a node owns it, Mesh.cpp:517-559; ghost node lists :767-795) and LinearEquation::InitPde
// (KKghost_nd, LinearEquation.cpp:239-280) and hands to VecCreateGhost (PetscVector.hpp:515-569).
// The only communication is one personalised all-to-all of 64-bit ids (every rank tells the owners which of their nodes it reads);
// the caller provides it (MPI_Alltoallv in an MPI launcher, torch.distributed / sockets in the Python harness).
#include "fh_internal.h"
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>

struct fh_dd_plan_s {
  int rank = 0, nranks = 1, n = 0;
  std::vector<int> owned, ghost, newid, send_idx, send_counts, recv_counts;
  // the reference's global numbering of the level: every rank owns one contiguous range (Mesh.cpp:735-741 _dofOffset, LinearEquation.cpp:
  // 212-237 KKoffset), ghosts are addressed by the owner's global index (KKghost_nd, :239-280)
  std::vector<int64_t> offsets;        // [nranks + 1] first global index of every rank
  std::vector<int64_t> ghost_global;   // [n_ghost] global index of every ghost, in ghost order
};

// box partition of the structured hierarchy (the METIS stand-in): rank (c0, c1, c2) of a p0 x p1 x p2 grid owns the unit
// cube [c, c + 1); a level-`level` mesh of nb coarse elements per unit has grid spacing 1 / (2 nb 2^level).  Global id = lexicographic
// index of the (exact, dyadic) coordinates; owner = the rank whose cube holds the node, the LOWER rank on shared faces.
extern "C" int fh_dd_box_node_keys(int n, const double* coords /* [n*3] */, int level, int nb, const int p[3], int64_t* gid, int* owner) {
  FH_REQUIRE(n >= 0 && coords && p && gid && owner && nb >= 1 && level >= 0 && p[0] >= 1 && p[1] >= 1 && p[2] >= 1, "fh_dd_box_node_keys: bad arguments");
  const int64_t S = (int64_t)2 * nb * ((int64_t)1 << level);
  const int64_t G0 = p[0] * S + 1, G1 = p[1] * S + 1;
  for (int i = 0; i < n; i++) {
    int64_t k[3];
    int oc[3];
    for (int d = 0; d < 3; d++) {
      k[d] = (int64_t)std::llround(coords[(size_t)i * 3 + d] * (double)S);
      FH_REQUIRE(k[d] >= 0 && k[d] <= p[d] * S, "fh_dd_box_node_keys: node %d lies outside the partitioned box", i);
      oc[d] = k[d] == 0 ? 0 : (int)((k[d] - 1) / S);     // a node on the face between two cubes belongs to the LOWER one (Mesh.cpp:517-559)
    }
    gid[i] = k[0] + G0 * (k[1] + G1 * k[2]);
    owner[i] = oc[0] + p[0] * (oc[1] + p[1] * oc[2]);
  }
  return 0;
}

extern "C" int fh_dd_plan_create(int rank, int nranks, int n, const int64_t* gid, const int* owner, const unsigned char* need,
                                 fh_dd_alltoallv_fn alltoallv, void* user, fh_dd_plan_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(out && n >= 0 && (n == 0 || (gid && owner && need)) && nranks >= 1 && rank >= 0 && rank < nranks, "fh_dd_plan_create: bad arguments");
  FH_REQUIRE(nranks == 1 || alltoallv, "fh_dd_plan_create: several ranks need the all-to-all function");
  std::unique_ptr<fh_dd_plan_s> P(new fh_dd_plan_s());
  P->rank = rank;
  P->nranks = nranks;
  P->n = n;
  for (int i = 0; i < n; i++) {
    FH_REQUIRE(owner[i] >= 0 && owner[i] < nranks, "fh_dd_plan_create: node %d has owner %d", i, owner[i]);
    if (owner[i] == rank) P->owned.push_back(i);                 // ascending local id = the local FEMuS order
    else if (need[i]) P->ghost.push_back(i);
  }
  std::sort(P->ghost.begin(), P->ghost.end(), [&](int a, int b) {   // by owner rank, then by global id
    return owner[a] != owner[b] ? owner[a] < owner[b] : gid[a] < gid[b];
  });
  P->newid.assign(n, -1);
  for (size_t k = 0; k < P->owned.size(); k++) P->newid[P->owned[k]] = (int)k;
  for (size_t k = 0; k < P->ghost.size(); k++) P->newid[P->ghost[k]] = (int)(P->owned.size() + k);
  P->recv_counts.assign(nranks, 0);
  for (int g : P->ghost) P->recv_counts[owner[g]]++;
  P->send_counts.assign(nranks, 0);
  if (nranks > 1) {
    // ask the owners: the global ids of my ghosts go out grouped by owner, the ids others need from me come back
    std::vector<int64_t> req(P->ghost.size());
    for (size_t k = 0; k < P->ghost.size(); k++) req[k] = gid[P->ghost[k]];
    FH_REQUIRE(alltoallv(user, nullptr, P->recv_counts.data(), nullptr, P->send_counts.data()) == 0, "fh_dd_plan_create: the count exchange failed");
    int64_t tot64 = 0;
    for (int r = 0; r < nranks; r++) {
      FH_REQUIRE(P->send_counts[r] >= 0, "fh_dd_plan_create: negative count from rank %d", r);
      tot64 += P->send_counts[r];
    }
    FH_REQUIRE(tot64 <= (int64_t)INT32_MAX, "fh_dd_plan_create: %lld requested entries do not fit the 32-bit send list", (long long)tot64);
    const int tot = (int)tot64;
    std::vector<int64_t> got(std::max(tot, 1));
    FH_REQUIRE(alltoallv(user, req.data(), P->recv_counts.data(), got.data(), P->send_counts.data()) == 0, "fh_dd_plan_create: the id exchange failed");
    std::vector<int> srt(P->owned.size());
    std::iota(srt.begin(), srt.end(), 0);
    std::sort(srt.begin(), srt.end(), [&](int a, int b) { return gid[P->owned[a]] < gid[P->owned[b]]; });
    P->send_idx.assign(tot, 0);
    // a rank that cannot serve a request does not leave here -- the others would wait for it in the next exchange: it goes on with
    // the size exchange, where a negative size tells every rank, and all of them return the error together
    long long bad_gid = 0;
    bool bad = false;
    for (int k = 0; k < tot; k++) {
      const int64_t g = got[k];
      auto it = std::lower_bound(srt.begin(), srt.end(), g, [&](int a, int64_t v) { return gid[P->owned[a]] < v; });
      if (it == srt.end() || gid[P->owned[*it]] != g) {
        if (!bad) bad_gid = (long long)g;
        bad = true;
        continue;
      }
      P->send_idx[k] = *it;                                      // position among the owned entries = index into the owned part of a vector
    }
    // global numbering: owned counts of all ranks (one more exchange of one number per pair), then every owner tells the requesters
    // the GLOBAL index (offset + position) of the nodes they asked for -- the reverse of the request exchange
    std::vector<int> ones(nranks, 1), got1(nranks, 0);
    std::vector<int64_t> mine(nranks, bad ? (int64_t)-1 : (int64_t)P->owned.size()), theirs(nranks, 0);
    FH_REQUIRE(alltoallv(user, nullptr, ones.data(), nullptr, got1.data()) == 0, "fh_dd_plan_create: the count exchange failed");
    FH_REQUIRE(alltoallv(user, mine.data(), ones.data(), theirs.data(), got1.data()) == 0, "fh_dd_plan_create: the size exchange failed");
    FH_REQUIRE(!bad, "fh_dd_plan_create: a requested node (global id %lld) is not owned by rank %d", bad_gid, rank);
    for (int r = 0; r < nranks; r++) FH_REQUIRE(theirs[r] >= 0, "fh_dd_plan_create: rank %d could not serve a ghost request (inconsistent ownership map)", r);
    P->offsets.assign(nranks + 1, 0);
    for (int r = 0; r < nranks; r++) P->offsets[r + 1] = P->offsets[r] + theirs[r];
    std::vector<int64_t> reply(std::max(tot, 1));
    for (int k = 0; k < tot; k++) reply[k] = P->offsets[rank] + P->send_idx[k];
    std::vector<int> rc2(nranks, 0);
    FH_REQUIRE(alltoallv(user, nullptr, P->send_counts.data(), nullptr, rc2.data()) == 0, "fh_dd_plan_create: the count exchange failed");
    for (int r = 0; r < nranks; r++) FH_REQUIRE(rc2[r] == P->recv_counts[r], "fh_dd_plan_create: rank %d answers %d ids, %d were asked for", r, rc2[r], P->recv_counts[r]);
    P->ghost_global.assign(std::max<size_t>(P->ghost.size(), 1), 0);
    FH_REQUIRE(alltoallv(user, reply.data(), P->send_counts.data(), P->ghost_global.data(), rc2.data()) == 0, "fh_dd_plan_create: the reply exchange failed");
    P->ghost_global.resize(P->ghost.size());
  } else {
    P->offsets = {0, (int64_t)P->owned.size()};
  }
  *out = P.release();
  return 0;
  FH_GUARD_END("fh_dd_plan_create")
}

// ---- system numbering of a multi-variable problem on several ranks (a9) ----------------------------------------------------------
// The reference numbers system rows rank by rank, and inside a rank variable by variable (LinearEquation::InitPde,
// LinearEquation.cpp:212-237):   KKoffset[0][0] = 0,  KKoffset[j][p] = KKoffset[j-1][p] + (own size of variable j-1 on rank p),
// KKoffset[0][p] = KKoffset[nvars][p-1];  KKIndex[j] = KKIndex[j-1] + (global size of variable j-1).
// dof_offset[k][p] = first mesh dof of variable k's family owned by rank p (Mesh::_dofOffset[solType], Mesh.cpp:735-741), [nvars][nranks+1].
extern "C" int fh_dd_system_offsets(int nvars, int nranks, const int64_t* dof_offset, int64_t* kk_offset /* [(nvars+1)][nranks] */,
                                    int64_t* kk_index /* [nvars+1], may be NULL */) {
  FH_REQUIRE(nvars >= 1 && nranks >= 1 && dof_offset && kk_offset, "fh_dd_system_offsets: bad arguments");
  for (int k = 0; k < nvars; k++)
    for (int p = 0; p < nranks; p++)
      FH_REQUIRE(dof_offset[(size_t)k * (nranks + 1) + p] <= dof_offset[(size_t)k * (nranks + 1) + p + 1],
                 "fh_dd_system_offsets: the dof offsets of variable %d decrease at rank %d", k, p);
  auto KK = [&](int j, int p) -> int64_t& { return kk_offset[(size_t)j * nranks + p]; };
  for (int p = 0; p < nranks; p++) {
    KK(0, p) = p == 0 ? 0 : KK(nvars, p - 1);
    for (int j = 1; j <= nvars; j++)
      KK(j, p) = KK(j - 1, p) + (dof_offset[(size_t)(j - 1) * (nranks + 1) + p + 1] - dof_offset[(size_t)(j - 1) * (nranks + 1) + p]);
  }
  if (kk_index) {
    kk_index[0] = 0;
    for (int j = 1; j <= nvars; j++) kk_index[j] = kk_index[j - 1] + dof_offset[(size_t)(j - 1) * (nranks + 1) + nranks];
  }
  return 0;
}

// LinearEquation::GetSystemDof (LinearEquation.cpp:76-85): mesh dof `idof` of variable `var` -> system row
// KKoffset[var][p] + idof - dof_offset[var][p], p = the rank owning the dof (Mesh::BisectionSearch_find_processor_of_dof, Mesh.cpp:1004-1018)
extern "C" int fh_dd_system_dofs(int nvars, int nranks, const int64_t* dof_offset, const int64_t* kk_offset, int var, int n, const int64_t* idof,
                                 int64_t* sysdof, int* owner /* may be NULL */) {
  FH_REQUIRE(nvars >= 1 && nranks >= 1 && dof_offset && kk_offset && var >= 0 && var < nvars && n >= 0 && (n == 0 || (idof && sysdof)),
             "fh_dd_system_dofs: bad arguments");
  const int64_t* off = dof_offset + (size_t)var * (nranks + 1);
  for (int i = 0; i < n; i++) {
    FH_REQUIRE(idof[i] >= off[0] && idof[i] < off[nranks], "fh_dd_system_dofs: dof %lld of variable %d lies outside [%lld, %lld)", (long long)idof[i], var,
               (long long)off[0], (long long)off[nranks]);
    const int p = (int)(std::upper_bound(off, off + nranks + 1, idof[i]) - off) - 1;   // off[p] <= idof < off[p + 1] (empty ranks skipped)
    sysdof[i] = kk_offset[(size_t)var * nranks + p] + idof[i] - off[p];
    if (owner) owner[i] = p;
  }
  return 0;
}

// global numbering of the level as the reference keeps it: offsets[nranks + 1] (rank r owns [offsets[r], offsets[r + 1])), and the
// global index of every ghost (the list a GHOSTED vector is initialised with, NumericVector::init(N, n_local, ghost, ...))
extern "C" int fh_dd_plan_global(fh_dd_plan_t P, int64_t* offsets, int64_t* ghost_global) {
  FH_REQUIRE(P, "fh_dd_plan_global: null plan");
  if (offsets) std::copy(P->offsets.begin(), P->offsets.end(), offsets);
  if (ghost_global) std::copy(P->ghost_global.begin(), P->ghost_global.end(), ghost_global);
  return 0;
}

extern "C" int fh_dd_plan_sizes(fh_dd_plan_t P, int* n_owned, int* n_ghost, int* n_send) {
  FH_REQUIRE(P, "fh_dd_plan_sizes: null plan");
  if (n_owned) *n_owned = (int)P->owned.size();
  if (n_ghost) *n_ghost = (int)P->ghost.size();
  if (n_send) *n_send = (int)P->send_idx.size();
  return 0;
}

extern "C" int fh_dd_plan_get(fh_dd_plan_t P, int* owned, int* ghost, int* newid, int* send_counts, int* send_idx, int* recv_counts) {
  FH_REQUIRE(P, "fh_dd_plan_get: null plan");
  if (owned) std::copy(P->owned.begin(), P->owned.end(), owned);
  if (ghost) std::copy(P->ghost.begin(), P->ghost.end(), ghost);
  if (newid) std::copy(P->newid.begin(), P->newid.end(), newid);
  if (send_counts) std::copy(P->send_counts.begin(), P->send_counts.end(), send_counts);
  if (send_idx) std::copy(P->send_idx.begin(), P->send_idx.end(), send_idx);
  if (recv_counts) std::copy(P->recv_counts.begin(), P->recv_counts.end(), recv_counts);
  return 0;
}

// the exchange plan of this level on the device: RCCL (id128 of rank 0 / a parent plan's communicator) or the host-staged transport
extern "C" int fh_dd_plan_halo(fh_dd_plan_t P, fh_ctx_t ctx, const char id128[128], fh_halo_t parent, fh_halo_t* halo) {
  FH_REQUIRE(P && ctx && halo, "fh_dd_plan_halo: null argument");
  if (parent) return fh_halo_create_shared(parent, P->send_counts.data(), P->send_idx.data(), P->recv_counts.data(), halo);
  return fh_halo_create(ctx, P->rank, P->nranks, id128, P->send_counts.data(), P->send_idx.data(), P->recv_counts.data(), halo);
}

extern "C" int fh_dd_plan_destroy(fh_dd_plan_t P) {
  delete P;
  return 0;
}
