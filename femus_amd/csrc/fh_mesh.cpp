// Host-side mesh and DOF-map layer (a8-a11, a13, a14 of SURVEY 8): box generator, uniform refinement,
// first-touch node numbering (nprocs = 1), boundary flags, CSR pattern, prolongator.  Integer work must be
// bit-exact with FEMuS; it is setup code and stays on the host exactly as in the reference.
//   MeshGeneration.cpp:790-849 (nodes), :979-1075 (elements + boundary flags)
//   MeshRefinement.cpp:240-294 (children), :356-417 (edge mid-points), :513-620 (face / element centres)
//   Mesh.cpp:517-559 (node renumbering), MeshRefinement.cpp:468-475 (fine coordinates = P x coarse)
//   LinearEquation.cpp:407-548 (sparsity), LinearImplicitSystem.cpp:761-909,1032-1120 (prolongator)
#include "fh_internal.h"
#include "fh_fe.h"
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <unordered_map>

struct fh_mesh_s {
  int geom = 0, dim = 3, nloc = 27, nel = 0, nnode = 0, level = 0;
  int own[3] = {0, 0, 0};
  std::vector<int> elem_dof;      // [nel*nloc]
  std::vector<double> coords;     // [nnode*dim]
  std::vector<int> face_flag;     // [nel*nfaces]
  std::vector<int> child;         // [nel*nchild] (set by refine on the coarse mesh; -1 padded for copied elements)
  std::vector<char> refined;      // [nel] set by refine on the coarse mesh: element was split
  std::vector<int> elem_level;    // [nel] refinement level of every element (Elem.hpp:372-374)
  std::vector<int> elem_group, elem_material;   // [nel] Gambit group / material number (Elem.hpp: GetElementGroup / GetElementMaterial), inherited by children; empty = group 1, material 2 ... as below
  bool homogeneous = true;        // Mesh::GetIfHomogeneous: no element of the father level was left unrefined
  // hanging-node constraints: 0 = the map exactly as Mesh::GetAMRRestrictionAndAMRSolidMark builds it (default), 1 = only the
  // description to the coarsest level for a node on two interfaces at once + full expansion of chains (rows sum to one);
  // inherited by refined meshes (fh_mesh_set_amr_mode)
  int amr_mode = 0;
  // hanging-node rows already computed for this mesh: [fe == 2], valid for amr_cache_mode (cleared when coordinates or the mode change)
  std::shared_ptr<struct AmrRows> amr_cache[2];
  int amr_cache_mode[2] = {-1, -1};
  // Dirichlet node list per family ([fe == 2]) once computed: the boundary flags of a mesh change only through fh_mesh_clear_boundary_faces
  mutable std::vector<int> dir_cache[4];      // per FE family 0 .. 3
  mutable bool dir_valid[4] = {false, false, false, false};
  // the same arrays in device memory (fh_mesh_refine_device, fh_mesh_device): dropped whenever a host array they mirror is rewritten
  fh_mesh_dev* dev = nullptr;
  ~fh_mesh_s() { fh_meshdev_free(dev); }
};

using namespace fhfe;

// Mesh.cpp:517-559 with nprocs = 1
static void first_touch_renumber(fh_mesh_s& m, int nnode) {
  const int nv = nvert_of(m.geom), ne = nedge_end_of(m.geom), nc = nloc_of(m.geom);
  std::vector<int> map(nnode, -1);
  int counter = 0;
  const int rng[4] = {0, nv, ne, nc};
  for (int k = 0; k < 3; k++) {
    for (int iel = 0; iel < m.nel; iel++)
      for (int i = rng[k]; i < rng[k + 1]; i++) {
        int ii = m.elem_dof[(size_t)iel * nc + i];
        if (map[ii] < 0) map[ii] = counter++;
      }
    m.own[k] = counter;
  }
  for (auto& v : m.elem_dof) v = map[v];
  if (!m.coords.empty()) {
    std::vector<double> c(m.coords.size());
    for (int i = 0; i < nnode; i++)
      for (int d = 0; d < m.dim; d++) c[(size_t)map[i] * m.dim + d] = m.coords[(size_t)i * m.dim + d];
    m.coords.swap(c);
  }
  m.nnode = nnode;
}

extern "C" int fh_mesh_box(int nx, int ny, int nz, const double lo[3], const double hi[3], fh_mesh_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(nx > 0 && ny > 0 && nz >= 0 && out, "fh_mesh_box: bad arguments");
  {   // 32-bit ids (the reference's, PetscVector.hpp:536): the node count and nel * 27 must fit
    const int64_t nn = (2ll * nx + 1) * (2ll * ny + 1) * (nz ? 2ll * nz + 1 : 1), ne = (int64_t)nx * ny * (nz ? nz : 1);
    FH_REQUIRE(nn < (1ll << 31) && ne * 27 < (1ll << 31), "fh_mesh_box: %lld nodes / %lld elements do not fit 32-bit ids", (long long)nn, (long long)ne);
  }
  std::unique_ptr<fh_mesh_s> holder(new fh_mesh_s());
  fh_mesh_s* m = holder.get();
  m->geom = (nz == 0) ? GEOM_QUAD : GEOM_HEX;
  m->dim = dim_of(m->geom);
  m->nloc = nloc_of(m->geom);
  const int n[3] = {nx, ny, nz == 0 ? 0 : nz};
  for (int d = 0; d < m->dim; d++) FH_REQUIRE(lo[d] < hi[d], "fh_mesh_box: lo >= hi");
  const int px = 2 * nx + 1, py = 2 * ny + 1, pz = (m->dim == 3) ? 2 * nz + 1 : 1;
  const int nnode = px * py * pz;
  m->coords.resize((size_t)nnode * m->dim);
  int id = 0;
  for (int k = 0; k < pz; k++)
    for (int j = 0; j < py; j++)
      for (int i = 0; i < px; i++, id++) {
        const int ijk[3] = {i, j, k};
        for (int d = 0; d < m->dim; d++)
          m->coords[(size_t)id * m->dim + d] = ((double)ijk[d] / (double)(2 * n[d])) * (hi[d] - lo[d]) + lo[d];
      }
  m->nel = nx * ny * (m->dim == 3 ? nz : 1);
  const int nf = nfaces_of(m->geom);
  m->elem_dof.resize((size_t)m->nel * m->nloc);
  m->face_flag.assign((size_t)m->nel * nf, -1);
  int iel = 0;
  for (int k = 0; k < (m->dim == 3 ? nz : 1); k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++, iel++) {
        for (int l = 0; l < m->nloc; l++) {
          int a = 2 * i + xc(m->geom, l, 0) + 1, b = 2 * j + xc(m->geom, l, 1) + 1;
          int c = (m->dim == 3) ? 2 * k + xc(m->geom, l, 2) + 1 : 0;
          m->elem_dof[(size_t)iel * m->nloc + l] = a + px * (b + c * py);
        }
        int* ff = &m->face_flag[(size_t)iel * nf];
        if (m->dim == 3) {
          if (k == 0) ff[4] = -2;
          if (k == nz - 1) ff[5] = -7;
          if (j == 0) ff[0] = -3;
          if (j == ny - 1) ff[2] = -5;
          if (i == 0) ff[3] = -6;
          if (i == nx - 1) ff[1] = -4;
        } else {
          if (j == 0) ff[0] = -2;
          if (i == nx - 1) ff[1] = -3;
          if (j == ny - 1) ff[2] = -4;
          if (i == 0) ff[3] = -5;
        }
      }
  m->elem_level.assign(m->nel, 0);
  first_touch_renumber(*m, nnode);
  *out = holder.release();
  return 0;
  FH_GUARD_END("fh_mesh_box")
}

struct Key3 {
  int a, b, c;
  bool operator==(const Key3& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct Key3Hash {
  size_t operator()(const Key3& k) const {
    uint64_t h = (uint64_t)k.a * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)k.b + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h ^= (uint64_t)k.c + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return (size_t)h;
  }
};

// open-addressing map (a, b, c) -> node id for the edge / face tables of a refinement: millions of look-ups, one allocation.  The slot index is
// a * S + (hash bits of b, c): a is the smallest vertex of the edge / face and vertex ids follow the element order (first-touch numbering),
// so consecutive elements probe neighbouring cache lines instead of the whole table.  S covers the keys one vertex can be the smallest of
// (6 edges, 12 faces on a hexahedral grid; the coarse vertices, numbered first and contiguously, reach that bound all at once -- fewer slots
// per vertex make their block overflow into one long probe chain)
struct PairMap {
  struct Slot {
    int a, b, c, value;
  };
  std::vector<Slot> slot;
  size_t cap = 0;
  int per = 8, shift = 61;
  PairMap(size_t expected, size_t n_first, int log2_per) : per(1 << log2_per), shift(64 - log2_per) {
    cap = std::max<size_t>(std::max(expected * 2, n_first * (size_t)per), 64);
    slot.assign(cap, Slot{-1, -1, -1, -1});
  }
  // the id stored for the key, or `fresh` after storing it (then *created = true)
  int find_or_insert(int a, int b, int c, int fresh, bool* created) {
    const uint64_t h = ((uint64_t)(uint32_t)b * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)(c + 1) * 0xC2B2AE3D27D4EB4Full);
    size_t i = ((size_t)(uint32_t)a * per + (size_t)(h >> shift)) % cap;
    for (;; i = (i + 1 == cap) ? 0 : i + 1) {
      Slot& s = slot[i];
      if (s.value < 0) {
        s = Slot{a, b, c, fresh};
        *created = true;
        return fresh;
      }
      if (s.a == a && s.b == b && s.c == c) {
        *created = false;
        return s.value;
      }
    }
  }
};

static void inherit_groups(const fh_mesh_s* mc, fh_mesh_s* m);

// MeshRefinement::RefineMesh (MeshRefinement.cpp:197-493) for nprocs = 1.  flags == NULL: every element is split (uniform
// level).  Otherwise elements of the current level with a nonzero flag are split and all the others are carried over
// unchanged (their node ids, boundary flags and level), which makes the new level non-homogeneous (AMR).
extern "C" int fh_mesh_refine_flagged(fh_mesh_t mc, const unsigned char* flags, fh_mesh_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(mc && out, "fh_mesh_refine: null argument");
  const int geom = mc->geom, dim = mc->dim, nc = mc->nloc;
  const int nv = nvert_of(geom), ne = nedge_end_of(geom), nch = nv, nf = nfaces_of(geom);
  FH_REQUIRE((int64_t)mc->nel * nch * nc < (1ll << 31), "fh_mesh_refine: the refinement of %d elements does not fit 32-bit ids", mc->nel);
  std::unique_ptr<fh_mesh_s> holder(new fh_mesh_s());
  fh_mesh_s* m = holder.get();
  m->geom = geom;
  m->dim = dim;
  m->nloc = nc;
  m->level = mc->level + 1;
  m->amr_mode = mc->amr_mode;
  if (mc->elem_level.empty()) mc->elem_level.assign(mc->nel, mc->level);
  fh_meshdev_free(mc->dev);     // its child table is about to change
  mc->dev = nullptr;
  // only elements of the current level can be refined (Elem.hpp:358-360)
  mc->refined.assign(mc->nel, 0);
  std::vector<int> start(mc->nel + 1, 0);
  for (int iel = 0; iel < mc->nel; iel++) {
    const bool r = (mc->elem_level[iel] == mc->level) && (!flags || flags[iel]);
    mc->refined[iel] = r;
    if (!r) m->homogeneous = false;
    start[iel + 1] = start[iel] + (r ? nch : 1);
  }
  m->nel = start[mc->nel];
  m->elem_dof.assign((size_t)m->nel * nc, -1);
  m->face_flag.assign((size_t)m->nel * nf, -1);
  m->elem_level.assign(m->nel, 0);
  // tables derived from the node coordinates
  int f2c[8][8];
  for (int j = 0; j < nch; j++)
    for (int v = 0; v < nv; v++) f2c[j][v] = fine2coarse_vertex(geom, j, v);
  int edge_v[12][2];
  for (int e = nv; e < ne; e++) {
    int cnt = 0;
    for (int v = 0; v < nv && cnt < 2; v++) {
      bool on = true;
      for (int d = 0; d < dim; d++)
        if (xc(geom, e, d) != 0 && xc(geom, e, d) != xc(geom, v, d)) on = false;
      if (on) edge_v[e - nv][cnt++] = v;
    }
  }
  int face_v[6][4];
  bool child_on_face[6][8];
  for (int f = 0; f < nf; f++) {
    const int centre = (geom == GEOM_HEX) ? 20 + f : 4 + f;
    int d0 = 0;
    for (int d = 0; d < dim; d++)
      if (xc(geom, centre, d) != 0) d0 = d;
    int cnt = 0;
    for (int v = 0; v < nv; v++) {
      bool on = xc(geom, v, d0) == xc(geom, centre, d0);
      child_on_face[f][v] = on;
      if (on && cnt < 4) face_v[f][cnt++] = v;
    }
  }
  // children: vertices and boundary flags (:240-278); copies of the elements that are not refined (:296-331)
  mc->child.assign((size_t)mc->nel * nch, -1);
  for (int iel = 0; iel < mc->nel; iel++) {
    if (mc->refined[iel]) {
      for (int j = 0; j < nch; j++) {
        const int jel = start[iel] + j;
        mc->child[(size_t)iel * nch + j] = jel;
        m->elem_level[jel] = mc->elem_level[iel] + 1;
        for (int v = 0; v < nv; v++) m->elem_dof[(size_t)jel * nc + v] = mc->elem_dof[(size_t)iel * nc + f2c[j][v]];
        for (int f = 0; f < nf; f++) {
          int value = mc->face_flag[(size_t)iel * nf + f];
          if (value < -1 && child_on_face[f][j]) m->face_flag[(size_t)jel * nf + f] = value;
        }
      }
    } else {
      const int jel = start[iel];
      mc->child[(size_t)iel * nch] = jel;
      m->elem_level[jel] = mc->elem_level[iel];
      for (int i = 0; i < nc; i++) m->elem_dof[(size_t)jel * nc + i] = mc->elem_dof[(size_t)iel * nc + i];
      for (int f = 0; f < nf; f++) {
        int value = mc->face_flag[(size_t)iel * nf + f];
        if (value < -1) m->face_flag[(size_t)jel * nf + f] = value;
      }
    }
  }
  auto fresh = [&](int iel) { return m->elem_level[iel] == m->level; };   // GetIfFatherHasBeenRefined
  int nnodes = mc->nnode;
  // edge mid-points (:356-417): first visit in (element, local edge) order creates the node
  {
    PairMap emap((size_t)m->nel * (size_t)(ne - nv) / 2 + 64, (size_t)mc->nnode, 3);
    for (int iel = 0; iel < m->nel; iel++) {
      if (!fresh(iel)) continue;
      for (int e = nv; e < ne; e++) {
        int a = m->elem_dof[(size_t)iel * nc + edge_v[e - nv][0]], b = m->elem_dof[(size_t)iel * nc + edge_v[e - nv][1]];
        if (a > b) std::swap(a, b);
        bool created;
        m->elem_dof[(size_t)iel * nc + e] = emap.find_or_insert(a, b, -1, nnodes, &created);
        if (created) nnodes++;
      }
    }
  }
  // quad-face centres of hexahedra (:526-561): (element, face 0..5) order
  if (geom == GEOM_HEX) {
    PairMap fmap((size_t)m->nel * 4 + 64, (size_t)mc->nnode, 4);
    for (int iel = 0; iel < m->nel; iel++) {
      if (!fresh(iel)) continue;
      for (int f = 0; f < 6; f++) {
        int v[4];
        for (int k = 0; k < 4; k++) v[k] = m->elem_dof[(size_t)iel * nc + face_v[f][k]];
        std::sort(v, v + 4);
        bool created;   // the three smallest vertices name the face
        m->elem_dof[(size_t)iel * nc + 20 + f] = fmap.find_or_insert(v[0], v[1], v[2], nnodes, &created);
        if (created) nnodes++;
      }
    }
  }
  // element centres (:598-616)
  for (int iel = 0; iel < m->nel; iel++)
    if (fresh(iel)) m->elem_dof[(size_t)iel * nc + nc - 1] = nnodes++;
  first_touch_renumber(*m, nnodes);
  // fine coordinates = biquadratic mesh prolongator x coarse coordinates (MeshRefinement.cpp:468-475);
  // rows are sums over the coarse element nodes in increasing global column order (CSR order)
  std::vector<double> EP;
  elem_prolongator(geom, FE_BIQUADRATIC, EP);
  m->coords.assign((size_t)m->nnode * dim, 0.0);
  // non-zero weights of every (child, node) row, once
  std::vector<int> nzn((size_t)nch * nc, 0);
  std::vector<unsigned char> nzmask((size_t)nch * nc * nc, 0);
  for (int ji = 0; ji < nch * nc; ji++)
    for (int k = 0; k < nc; k++) nzmask[(size_t)ji * nc + k] = EP[(size_t)ji * nc + k] != 0.0;
  // a row shared by several coarse elements gets the same sum from each of them (same weights on the same global columns, added in
  // increasing column order), so the coarse elements are split over threads and whoever claims a row first writes it
  std::unique_ptr<std::atomic<unsigned char>[]> done(new std::atomic<unsigned char>[(size_t)m->nnode]);
  for (int i = 0; i < m->nnode; i++) done[i].store(0, std::memory_order_relaxed);
  const int nth = (mc->nel >= 4096) ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
  auto work = [&](int e0, int e1) {
    int order[27];
    double xs[27 * 3];
    for (int iel = e0; iel < e1; iel++) {
      const int* cd = &mc->elem_dof[(size_t)iel * nc];
      if (!mc->refined[iel]) {
        const int jel = start[iel];
        for (int i = 0; i < nc; i++) {
          const int row = m->elem_dof[(size_t)jel * nc + i];
          if (done[row].exchange(1, std::memory_order_relaxed)) continue;
          for (int d = 0; d < dim; d++) m->coords[(size_t)row * dim + d] = mc->coords[(size_t)cd[i] * dim + d];
        }
        continue;
      }
      for (int k = 0; k < nc; k++) order[k] = k;
      std::sort(order, order + nc, [&](int a, int b) { return cd[a] < cd[b]; });
      for (int kk = 0; kk < nc; kk++)
        for (int d = 0; d < dim; d++) xs[kk * 3 + d] = mc->coords[(size_t)cd[order[kk]] * dim + d];
      for (int j = 0; j < nch; j++) {
        const int jel = start[iel] + j;
        for (int i = 0; i < nc; i++) {
          const int row = m->elem_dof[(size_t)jel * nc + i];
          if (done[row].exchange(1, std::memory_order_relaxed)) continue;
          const double* pr = &EP[((size_t)j * nc + i) * nc];
          const unsigned char* nz = &nzmask[((size_t)j * nc + i) * nc];
          double sum[3] = {0.0, 0.0, 0.0};
          for (int kk = 0; kk < nc; kk++) {
            const int k = order[kk];
            if (!nz[k]) continue;
            for (int d = 0; d < dim; d++) sum[d] += pr[k] * xs[kk * 3 + d];
          }
          for (int d = 0; d < dim; d++) m->coords[(size_t)row * dim + d] = sum[d];
        }
      }
    }
  };
  if (nth == 1) {
    work(0, mc->nel);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nth; t++) th.emplace_back(work, (int)((int64_t)mc->nel * t / nth), (int)((int64_t)mc->nel * (t + 1) / nth));
    for (auto& x : th) x.join();
  }
  inherit_groups(mc, m);
  *out = holder.release();
  return 0;
  FH_GUARD_END("fh_mesh_refine_flagged")
}

// children carry the group and material of their father (MeshRefinement.cpp:263-266: SetElementGroup / SetElementMaterial of the new element)
static void inherit_groups(const fh_mesh_s* mc, fh_mesh_s* m) {
  if (mc->elem_group.empty()) return;
  const int nch = nvert_of(mc->geom);
  m->elem_group.assign(m->nel, 0);
  m->elem_material.assign(m->nel, 0);
  for (int iel = 0; iel < mc->nel; iel++)
    for (int j = 0; j < nch; j++) {
      const int jel = mc->child[(size_t)iel * nch + j];
      if (jel < 0) continue;
      m->elem_group[jel] = mc->elem_group[iel];
      m->elem_material[jel] = mc->elem_material[iel];
    }
}

// Elem::GetElementGroup / GetElementMaterial per element; a generated box has one group (1) of fluid material (2), as MeshGeneration leaves it
extern "C" int fh_mesh_elem_groups(fh_mesh_t m, int* group, int* material) {
  FH_REQUIRE(m, "fh_mesh_elem_groups: null mesh");
  for (int i = 0; i < m->nel; i++) {
    if (group) group[i] = m->elem_group.empty() ? 1 : m->elem_group[i];
    if (material) material[i] = m->elem_material.empty() ? 2 : m->elem_material[i];
  }
  return 0;
}

extern "C" int fh_mesh_refine(fh_mesh_t mc, fh_mesh_t* out) {
  FH_GUARD_BEGIN return fh_mesh_refine_flagged(mc, nullptr, out);   FH_GUARD_END("fh_mesh_refine")
}

// reference-element tables of a refinement, derived from the node coordinates of the reference element as fh_mesh_refine_flagged derives them
static void refine_tables(int geom, fh_refine_tables& T) {
  const int dim = dim_of(geom);
  T.nv = nvert_of(geom); T.ne = nedge_end_of(geom); T.nc = nloc_of(geom); T.nch = T.nv; T.nf = nfaces_of(geom); T.dim = dim;
  memset(T.f2c, 0, sizeof(T.f2c)); memset(T.edge_v, 0, sizeof(T.edge_v)); memset(T.face_v, 0, sizeof(T.face_v));
  memset(T.face_diag, 0, sizeof(T.face_diag)); memset(T.cof, 0, sizeof(T.cof));
  for (int j = 0; j < T.nch; j++)
    for (int v = 0; v < T.nv; v++) T.f2c[j][v] = fine2coarse_vertex(geom, j, v);
  for (int e = T.nv; e < T.ne; e++) {
    int cnt = 0;
    for (int v = 0; v < T.nv && cnt < 2; v++) {
      bool on = true;
      for (int d = 0; d < dim; d++)
        if (xc(geom, e, d) != 0 && xc(geom, e, d) != xc(geom, v, d)) on = false;
      if (on) T.edge_v[e - T.nv][cnt++] = v;
    }
  }
  for (int f = 0; f < T.nf; f++) {
    const int centre = (geom == GEOM_HEX) ? 20 + f : 4 + f;
    int d0 = 0;
    for (int d = 0; d < dim; d++)
      if (xc(geom, centre, d) != 0) d0 = d;
    int cnt = 0;
    for (int v = 0; v < T.nv; v++) {
      const bool on = xc(geom, v, d0) == xc(geom, centre, d0);
      T.cof[f][v] = on;
      if (on && cnt < 4) T.face_v[f][cnt++] = v;
    }
    if (geom == GEOM_HEX)          // the vertex of the face that shares no edge with vertex k: both in-face coordinates differ
      for (int k = 0; k < 4; k++)
        for (int k2 = 0; k2 < 4; k2++) {
          int differ = 0;
          for (int d = 0; d < dim; d++) differ += xc(geom, T.face_v[f][k], d) != xc(geom, T.face_v[f][k2], d);
          if (differ == 2) T.face_diag[f][k] = k2;
        }
  }
  elem_prolongator(geom, FE_BIQUADRATIC, T.EP);
  T.cnt.assign((size_t)T.nch * T.nc, 0);
  T.nzk.assign((size_t)T.nch * T.nc * T.nc, 0);
  for (int ji = 0; ji < T.nch * T.nc; ji++)
    for (int k = 0; k < T.nc; k++)
      if (T.EP[(size_t)ji * T.nc + k] != 0.0) T.nzk[(size_t)ji * T.nc + T.cnt[ji]++] = k;
}

int fh_mesh_device(fh_ctx_t ctx, fh_mesh_t m, fh_mesh_dev** dev) {
  FH_REQUIRE(ctx && m && dev, "fh_mesh_device: null argument");
  if (m->dev && m->dev->ctx != ctx) {
    fh_meshdev_free(m->dev);
    m->dev = nullptr;
  }
  if (!m->dev) {
    if (m->elem_level.empty()) m->elem_level.assign(m->nel, m->level);
    FH_TRY(fh_meshdev_upload(ctx, m->nel, m->nnode, m->nloc, m->dim, nfaces_of(m->geom), m->elem_dof.data(), m->coords.data(), m->face_flag.data(),
                             m->elem_level.data(), &m->dev));
  }
  *dev = m->dev;
  return 0;
}

// The same refinement on the device (fh_meshdev.hip): identical arrays, bit for bit; the new mesh keeps its device copy for the set-up calls
// that follow (fh_mat_create_from_mesh, fh_assembler_create_mesh, fh_build_prolongator) and a host copy for everything else.
extern "C" int fh_mesh_refine_device(fh_ctx_t ctx, fh_mesh_t mc, const unsigned char* flags, fh_mesh_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && mc && out, "fh_mesh_refine_device: null argument");
  const int geom = mc->geom, nc = mc->nloc, nch = nvert_of(geom);
  FH_REQUIRE((int64_t)mc->nel * nch * nc < (1ll << 31), "fh_mesh_refine_device: the refinement of %d elements does not fit 32-bit ids", mc->nel);
  fh_mesh_dev* cdev = nullptr;
  FH_TRY(fh_mesh_device(ctx, mc, &cdev));
  fh_refine_tables T;
  refine_tables(geom, T);
  fh_refine_result R;
  FH_TRY(fh_meshdev_refine(ctx, T, cdev, mc->level, flags, &R));
  std::unique_ptr<fh_mesh_s> holder(new fh_mesh_s());
  fh_mesh_s* m = holder.get();
  m->dev = R.dev;
  m->geom = geom;
  m->dim = mc->dim;
  m->nloc = nc;
  m->level = mc->level + 1;
  m->amr_mode = mc->amr_mode;
  m->nel = R.nel;
  m->nnode = R.nnode;
  for (int k = 0; k < 3; k++) m->own[k] = R.own[k];
  m->elem_dof.swap(R.elem_dof);
  m->coords.swap(R.coords);
  m->face_flag.swap(R.face_flag);
  m->elem_level.swap(R.elem_level);
  mc->child.swap(R.child);
  mc->refined.swap(R.refined);
  m->homogeneous = m->nel == mc->nel * nch;
  inherit_groups(mc, m);
  *out = holder.release();
  return 0;
  FH_GUARD_END("fh_mesh_refine_device")
}

// MeshRefinement::FlagElementsToRefine (:88-101): the flag function is evaluated at the mean of the element vertices
extern "C" int fh_mesh_elem_centroids(fh_mesh_t m, double* xc3) {
  FH_REQUIRE(m && xc3, "fh_mesh_elem_centroids: null argument");
  const int nv = nvert_of(m->geom);
  for (int iel = 0; iel < m->nel; iel++) {
    double x[3] = {0, 0, 0};
    for (int v = 0; v < nv; v++)
      for (int d = 0; d < m->dim; d++) x[d] += m->coords[(size_t)m->elem_dof[(size_t)iel * m->nloc + v] * m->dim + d];
    for (int d = 0; d < 3; d++) xc3[(size_t)iel * 3 + d] = x[d] / nv;
  }
  return 0;
}

extern "C" int fh_mesh_elem_levels(fh_mesh_t m, int* levels, int* homogeneous) {
  FH_REQUIRE(m, "fh_mesh_elem_levels: null argument");
  if (levels)
    for (int iel = 0; iel < m->nel; iel++) levels[iel] = m->elem_level.empty() ? m->level : m->elem_level[iel];
  if (homogeneous) *homogeneous = m->homogeneous ? 1 : 0;
  return 0;
}

extern "C" int fh_mesh_clear_boundary_faces(fh_mesh_t m, unsigned face_mask) {
  const int nf = nfaces_of(m->geom);
  for (int iel = 0; iel < m->nel; iel++)
    for (int f = 0; f < nf; f++)
      if ((face_mask >> f) & 1u) m->face_flag[(size_t)iel * nf + f] = -1;
  m->amr_cache[0].reset();     // interface faces of the hanging-node search are the faces flagged -1: rows cached before are stale
  m->amr_cache[1].reset();
  for (bool& v : m->dir_valid) v = false;
  fh_meshdev_free(m->dev);
  m->dev = nullptr;
  return 0;
}

extern "C" int fh_mesh_set_coords(fh_mesh_t m, const double* coords) {
  FH_REQUIRE(m && coords, "fh_mesh_set_coords: null argument");
  memcpy(m->coords.data(), coords, m->coords.size() * sizeof(double));
  m->amr_cache[0].reset();     // the hanging-node weights are found through the coordinates
  m->amr_cache[1].reset();
  fh_meshdev_free(m->dev);
  m->dev = nullptr;
  return 0;
}

extern "C" int fh_mesh_destroy(fh_mesh_t m) {
  delete m;
  return 0;
}

extern "C" int fh_mesh_info(fh_mesh_t m, int* dim, int* nel, int* nnode, int* nloc, int own[3], int* level) {
  if (dim) *dim = m->dim;
  if (nel) *nel = m->nel;
  if (nnode) *nnode = m->nnode;
  if (nloc) *nloc = m->nloc;
  if (own) memcpy(own, m->own, 3 * sizeof(int));
  if (level) *level = m->level;
  return 0;
}

// host arrays of a mesh for the other translation units of the library (fh_io.cpp)
int fh_mesh_host_arrays(fh_mesh_t m, int* dim, int* geom, int* nel, int* nnode, int* nloc, int* n_linear, const int** elem_dof, const double** coords) {
  FH_REQUIRE(m, "null mesh");
  *dim = m->dim;
  *geom = m->geom;
  *nel = m->nel;
  *nnode = m->nnode;
  *nloc = m->nloc;
  *n_linear = m->own[0];
  *elem_dof = m->elem_dof.data();
  *coords = m->coords.data();
  return 0;
}

extern "C" int fh_mesh_get(fh_mesh_t m, int* elem_dof, double* coords, int* face_flag) {
  if (elem_dof) fh_copy_out(elem_dof, m->elem_dof);
  if (coords) fh_copy_out(coords, m->coords);
  if (face_flag) fh_copy_out(face_flag, m->face_flag);
  return 0;
}

extern "C" int fh_mesh_child_elems(fh_mesh_t m, int* child) {
  FH_REQUIRE(!m->child.empty(), "fh_mesh_child_elems: mesh has not been refined");
  fh_copy_out(child, m->child);
  return 0;
}

// dofs of a scalar variable of the family on this mesh (one process): Mesh::GetSolutionDof (Mesh.cpp:1021-1074) maps local node i of an element to the mesh node
// for the three Lagrange families -- nodes are numbered vertices, edge mid-points, the rest, so the linear and the serendipity family own the leading
// own[0] / own[1] ids (:1026-1050) -- and to the element for the piecewise constant one (:1056-1059)
static int mesh_ndofs(const fh_mesh_s* m, int fe) { return fe == FE_LINEAR ? m->own[0] : fe == FE_SERENDIPITY ? m->own[1] : fe == FE_CONSTANT ? m->nel : m->nnode; }

static void dirichlet_list(const fh_mesh_s* m, int fe, std::vector<int>& out) {
  const int slot = fe;
  if (fe == FE_CONSTANT) {          // element-owned: no dof lies on a face
    out.clear();
    return;
  }
  if (m->dir_valid[slot]) {
    out = m->dir_cache[slot];
    return;
  }
  const int nc = ndofs_of(m->geom, fe), nf = nfaces_of(m->geom), nl = m->nloc;
  std::vector<char> mark(mesh_ndofs(m, fe), 0);
  for (int f = 0; f < nf; f++) {
    const int centre = (m->geom == GEOM_HEX) ? 20 + f : 4 + f;
    int d0 = 0;
    for (int d = 0; d < m->dim; d++)
      if (xc(m->geom, centre, d) != 0) d0 = d;
    for (int iel = 0; iel < m->nel; iel++) {
      if (m->face_flag[(size_t)iel * nf + f] >= -1) continue;
      for (int i = 0; i < nc; i++)
        if (xc(m->geom, i, d0) == xc(m->geom, centre, d0)) mark[m->elem_dof[(size_t)iel * nl + i]] = 1;
    }
  }
  out.clear();
  for (size_t i = 0; i < mark.size(); i++)
    if (mark[i]) out.push_back((int)i);
  m->dir_cache[slot] = out;
  m->dir_valid[slot] = true;
}

extern "C" int fh_mesh_dirichlet_dofs(fh_mesh_t m, int fe, int* n, int* dofs) {
  FH_REQUIRE(fe_known(fe), "fh_mesh_dirichlet_dofs: fe must be 0 .. 3");
  std::vector<int> list;
  dirichlet_list(m, fe, list);
  FH_REQUIRE(*n >= (int)list.size(), "fh_mesh_dirichlet_dofs: capacity %d < %d", *n, (int)list.size());
  *n = (int)list.size();
  fh_copy_out(dofs, list);
  return 0;
}

// a11: union of element couplings per row, sorted
extern "C" int fh_pattern_from_elements(int nel, int nloc, const int* elem_dof, int ndof, int* rowptr, int* col) {
  FH_GUARD_BEGIN
  FH_REQUIRE(nel >= 0 && nloc > 0 && ndof >= 0 && rowptr, "fh_pattern_from_elements: bad arguments");
  std::vector<int> cnt(ndof + 1, 0);
  for (size_t k = 0; k < (size_t)nel * nloc; k++) {
    FH_REQUIRE(elem_dof[k] >= 0 && elem_dof[k] < ndof, "fh_pattern_from_elements: dof %d out of range", elem_dof[k]);
    cnt[elem_dof[k] + 1]++;
  }
  for (int i = 0; i < ndof; i++) cnt[i + 1] += cnt[i];
  std::vector<int> adj(cnt[ndof]), cur(cnt.begin(), cnt.end() - 1);
  for (int e = 0; e < nel; e++)
    for (int l = 0; l < nloc; l++) adj[cur[elem_dof[(size_t)e * nloc + l]]++] = e;
  // rows are independent: the sort/unique of every row's element dofs runs on all host cores (the 128^3 level has 1.08e9 entries)
  const unsigned nthreads = std::max(1u, std::min(std::thread::hardware_concurrency(), ndof > 65536 ? 32u : 1u));
  auto row_entries = [&](int r, std::vector<int>& buf) {
    buf.clear();
    for (int k = cnt[r]; k < cnt[r + 1]; k++) {
      const int* ed = elem_dof + (size_t)adj[k] * nloc;
      buf.insert(buf.end(), ed, ed + nloc);
    }
    std::sort(buf.begin(), buf.end());
    buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
  };
  auto parallel_rows = [&](auto&& body) {
    std::vector<std::thread> pool;
    const int chunk = (ndof + (int)nthreads - 1) / (int)nthreads;
    for (unsigned t = 0; t < nthreads; t++) {
      const int r0 = std::min(ndof, (int)t * chunk), r1 = std::min(ndof, r0 + chunk);
      if (r0 < r1) pool.emplace_back([&, r0, r1] { std::vector<int> buf; for (int r = r0; r < r1; r++) body(r, buf); });
    }
    for (auto& th : pool) th.join();
  };
  if (!col) {
    std::vector<int> len(ndof, 0);
    parallel_rows([&](int r, std::vector<int>& buf) { row_entries(r, buf); len[r] = (int)buf.size(); });
    int64_t total = 0;
    rowptr[0] = 0;
    for (int r = 0; r < ndof; r++) {
      total += len[r];
      FH_REQUIRE(total < 2147483647ll, "fh_pattern_from_elements: nnz overflows int32");
      rowptr[r + 1] = (int)total;
    }
  } else {
    std::atomic<int> bad(0);
    parallel_rows([&](int r, std::vector<int>& buf) {
      row_entries(r, buf);
      if (rowptr[r + 1] - rowptr[r] != (int)buf.size() || rowptr[r] < 0) { bad = 1; return; }
      std::copy(buf.begin(), buf.end(), col + rowptr[r]);
    });
    FH_REQUIRE(!bad && rowptr[0] == 0, "fh_pattern_from_elements: rowptr does not match (call with col=NULL first)");
  }
  return 0;
  FH_GUARD_END("fh_pattern_from_elements")
}

// the device builder lives in fh_setup.hip (this file stays plain host C++: tests/asan_host.sh); it gets the host arrays
int fh_prolongator_device(fh_ctx_t ctx, int nl, int nc, int nch, int nel_c, const int* child, const char* refined, const int* c_ed, size_t n_fed, const int* f_ed,
                          int nf, int ncc, const std::vector<double>& EP, const char* bf, const char* bc, fh_mat_t* out, const fh_mesh_dev* cdev, const fh_mesh_dev* fdev);

static int build_prolongator_device(fh_ctx_t ctx, fh_mesh_t mc, fh_mesh_t mf, int fe, int zero_bdc, fh_mat_t* out) {
  const int geom = mc->geom, nl = mc->nloc, nc = ndofs_of(geom, fe), nch = nvert_of(geom);
  const int nf = mesh_ndofs(mf, fe), ncc = mesh_ndofs(mc, fe);
  std::vector<double> EP;
  elem_prolongator(geom, fe, EP);
  std::vector<char> bf, bc;
  if (zero_bdc) {
    std::vector<int> lf, lc;
    dirichlet_list(mf, fe, lf);
    dirichlet_list(mc, fe, lc);
    bf.assign(nf, 0);
    bc.assign(ncc, 0);
    for (int r : lf) bf[r] = 1;
    for (int c : lc) bc[c] = 1;
  }
  return fh_prolongator_device(ctx, nl, nc, nch, mc->nel, mc->child.data(), mc->refined.data(), mc->elem_dof.data(), mf->elem_dof.size(), mf->elem_dof.data(), nf, ncc,
                               EP, zero_bdc ? bf.data() : nullptr, zero_bdc ? bc.data() : nullptr, out,
                               // device copies made by fh_mesh_refine_device on this context: nothing to upload but the boundary marks
                               (mc->dev && mf->dev && mc->dev->ctx == ctx && mf->dev->ctx == ctx && mc->dev->d_child) ? mc->dev : nullptr, mf->dev);
}

// a14: P (fine x coarse), INSERT semantics (first insert wins; duplicates are identical rows).  Elements that were not
// refined contribute identity rows (LinearImplicitSystem.cpp:796-806).
extern "C" int fh_build_prolongator(fh_ctx_t ctx, fh_mesh_t mc, fh_mesh_t mf, int fe, int zero_bdc, fh_mat_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && mc && mf && out, "fh_build_prolongator: null argument");
  FH_REQUIRE(fe_known(fe), "fh_build_prolongator: fe must be 0 (linear), 1 (serendipity), 2 (biquadratic) or 3 (piecewise constant)");
  const int geom = mc->geom, nl = mc->nloc, nc = ndofs_of(geom, fe), nch = nvert_of(geom);
  FH_REQUIRE(!mc->child.empty() && (int)mc->refined.size() == mc->nel, "fh_build_prolongator: fine is not the refinement of coarse");
  {
    int expect = 0;
    for (int iel = 0; iel < mc->nel; iel++) expect += mc->refined[iel] ? nch : 1;
    FH_REQUIRE(expect == mf->nel, "fh_build_prolongator: fine is not the refinement of coarse");
  }
  if (fe == FE_CONSTANT) {
    // quad0 / hex0 (set_prolongation_OneElement_All_FE with the one function 1 at the centres of the children, ElemType.cpp:439-532): every child -- and an
    // element carried over unrefined -- takes the value of its father
    std::vector<int> rowptr(mf->nel + 1), col(mf->nel, -1);
    std::vector<double> val(mf->nel, 1.0);
    for (int r = 0; r <= mf->nel; r++) rowptr[r] = r;
    for (int iel = 0; iel < mc->nel; iel++)
      for (int j = 0; j < (mc->refined[iel] ? nch : 1); j++) {
        const int jel = mc->child[(size_t)iel * nch + j];
        FH_REQUIRE(jel >= 0 && jel < mf->nel, "fh_build_prolongator: child element %d out of range", jel);
        col[jel] = iel;
      }
    for (int jel = 0; jel < mf->nel; jel++) FH_REQUIRE(col[jel] >= 0, "fh_build_prolongator: fine element %d has no father", jel);
    return fh_mat_create_csr(ctx, mf->nel, mc->nel, rowptr.data(), col.data(), val.data(), out);
  }
  if (ctx->device_setup) return build_prolongator_device(ctx, mc, mf, fe, zero_bdc, out);
  const int nf = mesh_ndofs(mf, fe), ncc = mesh_ndofs(mc, fe);
  std::vector<double> EP;
  elem_prolongator(geom, fe, EP);
  std::vector<int> rowptr(nf + 1, 0);
  std::vector<char> done(nf, 0);
  // pass 1: row lengths
  for (int iel = 0; iel < mc->nel; iel++) {
    const int nj = mc->refined[iel] ? nch : 1;
    for (int j = 0; j < nj; j++) {
      const int jel = mc->child[(size_t)iel * nch + j];
      for (int i = 0; i < nc; i++) {
        const int row = mf->elem_dof[(size_t)jel * nl + i];
        if (done[row]) continue;
        done[row] = 1;
        int cntr = 1;
        if (mc->refined[iel]) {
          cntr = 0;
          for (int k = 0; k < nc; k++) cntr += (EP[((size_t)j * nc + i) * nc + k] != 0.0);
        }
        rowptr[row + 1] = cntr;
      }
    }
  }
  for (int r = 0; r < nf; r++) rowptr[r + 1] += rowptr[r];
  std::vector<int> col(rowptr[nf]);
  std::vector<double> val(rowptr[nf]);
  std::fill(done.begin(), done.end(), 0);
  std::vector<char> bf, bc;
  if (zero_bdc) {
    std::vector<int> lf, lc;
    dirichlet_list(mf, fe, lf);
    dirichlet_list(mc, fe, lc);
    bf.assign(nf, 0);
    bc.assign(ncc, 0);
    for (int r : lf) bf[r] = 1;
    for (int c : lc) bc[c] = 1;
  }
  std::vector<std::pair<int, double>> rowbuf;
  for (int iel = 0; iel < mc->nel; iel++) {
    const int nj = mc->refined[iel] ? nch : 1;
    for (int j = 0; j < nj; j++) {
      const int jel = mc->child[(size_t)iel * nch + j];
      for (int i = 0; i < nc; i++) {
        const int row = mf->elem_dof[(size_t)jel * nl + i];
        if (done[row]) continue;
        done[row] = 1;
        rowbuf.clear();
        if (mc->refined[iel]) {
          for (int k = 0; k < nc; k++) {
            double v = EP[((size_t)j * nc + i) * nc + k];
            if (v == 0.0) continue;
            rowbuf.emplace_back(mc->elem_dof[(size_t)iel * nl + k], v);
          }
        } else {
          rowbuf.emplace_back(mc->elem_dof[(size_t)iel * nl + i], 1.0);
        }
        std::sort(rowbuf.begin(), rowbuf.end());
        int p = rowptr[row];
        for (auto& cv : rowbuf) {
          col[p] = cv.first;
          // pattern kept, value zeroed (mat_zero_rows keeps the pattern)
          val[p++] = (zero_bdc && (bf[row] || bc[cv.first])) ? 0.0 : cv.second;
        }
      }
    }
  }
  return fh_mat_create_csr(ctx, nf, ncc, rowptr.data(), col.data(), val.data(), out);
  FH_GUARD_END("fh_build_prolongator")
}

// ---------------------------------------------------------------------------------------------------------------------
// a22: hanging-node constraints of a non-homogeneous level (Mesh::GetAMRRestrictionAndAMRSolidMark, Mesh.cpp:1352-1830)
// and the projection matrix built from them (LinearImplicitSystem::BuildAmrProlongatorMatrix, :912-1028).
//
// A face is an AMR interface when no other element shares all its vertices and it carries no boundary flag (near-face
// index -1).  For every pair of levels (coarse Lc < fine Lf): every node of a fine interface face that lies inside a
// coarse interface element without being one of its nodes hangs on that element; its weights are the coarse basis
// functions of the element's interface-face nodes at that point (|w| >= 1e-10).  Masters that hang themselves are
// resolved recursively.  A node on the interfaces with two coarser levels at once (3-D edges with a level jump of two)
// has two mathematically identical descriptions; the one to the coarsest level is kept.
// ---------------------------------------------------------------------------------------------------------------------
struct AmrRows {
  std::vector<int> hang;                 // sorted hanging dofs
  std::vector<int> ptr;                  // CSR over hang
  std::vector<int> master;
  std::vector<double> w;
};

static bool inverse_map_q2(int geom, int dim, const double* xv /* [nloc*dim] */, const double* xp, double* xi) {
  const int nl = nloc_of(geom);
  double phi[27], dphi[81];
  double scale = 1.0;
  for (int k = 0; k < nl * dim; k++) scale = std::max(scale, std::fabs(xv[k]) + 1.0);
  for (int d = 0; d < dim; d++) xi[d] = 0.0;
  for (int it = 0; it < 30; it++) {
    eval_basis(geom, FE_BIQUADRATIC, xi, phi, dphi);
    double r[3] = {0, 0, 0}, J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // J[b][a] = d x_b / d xi_a
    for (int j = 0; j < nl; j++)
      for (int b = 0; b < dim; b++) {
        r[b] += phi[j] * xv[j * dim + b];
        for (int a = 0; a < dim; a++) J[b][a] += dphi[j * dim + a] * xv[j * dim + b];
      }
    for (int b = 0; b < dim; b++) r[b] -= xp[b];
    double dx[3] = {0, 0, 0};
    if (dim == 2) {
      const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      if (det == 0.0) return false;
      dx[0] = (J[1][1] * r[0] - J[0][1] * r[1]) / det;
      dx[1] = (-J[1][0] * r[0] + J[0][0] * r[1]) / det;
    } else {
      const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                   c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
      if (det == 0.0) return false;
      const double inv[3][3] = {
          {c00 / det, (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det, (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det},
          {c01 / det, (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det, (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det},
          {c02 / det, (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det, (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det}};
      for (int a = 0; a < 3; a++) dx[a] = inv[a][0] * r[0] + inv[a][1] * r[1] + inv[a][2] * r[2];
    }
    double mx = 0.0;
    for (int a = 0; a < dim; a++) {
      xi[a] -= dx[a];
      mx = std::max(mx, std::fabs(dx[a]));
    }
    if (mx < 1e-14 * scale) return true;
  }
  return true;
}

static void amr_constraints_compute(const fh_mesh_s* m, int fe, AmrRows& out);
static void amr_constraints(fh_mesh_s* m, int fe, AmrRows& out) {
  const int k = fe == FE_BIQUADRATIC ? 1 : 0;
  if (!m->amr_cache[k] || m->amr_cache_mode[k] != m->amr_mode) {
    m->amr_cache[k] = std::make_shared<AmrRows>();
    amr_constraints_compute(m, fe, *m->amr_cache[k]);
    m->amr_cache_mode[k] = m->amr_mode;
  }
  out = *m->amr_cache[k];
}

static void amr_constraints_compute(const fh_mesh_s* m, int fe, AmrRows& out) {
  const int geom = m->geom, dim = m->dim, nl = m->nloc, nv = nvert_of(geom), nf = nfaces_of(geom), nc = ndofs_of(geom, fe);
  const int nfv = (dim == 3) ? 4 : 2;
  out = AmrRows();
  out.ptr.push_back(0);
  if (m->homogeneous || m->elem_level.empty()) return;
  // face vertices / face nodes from the local coordinates
  int face_v[6][4], face_n[6][9], face_nn[6];
  for (int f = 0; f < nf; f++) {
    const int centre = (geom == GEOM_HEX) ? 20 + f : 4 + f;
    int d0 = 0;
    for (int d = 0; d < dim; d++)
      if (xc(geom, centre, d) != 0) d0 = d;
    int cv = 0, cn = 0;
    for (int i = 0; i < nl; i++)
      if (xc(geom, i, d0) == xc(geom, centre, d0)) {
        if (i < nv) face_v[f][cv++] = i;
        if (i < nc) face_n[f][cn++] = i;
      }
    face_nn[f] = cn;
  }
  // interface faces: vertex-set key seen exactly once and no boundary flag
  // (table: face key -> index into `seen`; the key's first entry is the smallest vertex, see PairMap)
  PairMap ftab((size_t)m->nel * (size_t)nf / 2 + 64, (size_t)m->own[0] + 1, dim == 3 ? 4 : 3);
  std::vector<int> seen;
  seen.reserve((size_t)m->nel * nf / 2 + 64);
  std::vector<int> fslot((size_t)m->nel * nf);
  for (int iel = 0; iel < m->nel; iel++)
    for (int f = 0; f < nf; f++) {
      int v[4] = {-1, -1, -1, -1};
      for (int k = 0; k < nfv; k++) v[k] = m->elem_dof[(size_t)iel * nl + face_v[f][k]];
      std::sort(v, v + nfv);
      bool created;
      const int id = ftab.find_or_insert(v[0], v[1], dim == 3 ? v[2] : -1, (int)seen.size(), &created);
      if (created) seen.push_back(0);
      seen[id]++;
      fslot[(size_t)iel * nf + f] = id;
    }
  struct IfaceElem {
    int iel;
    std::vector<int> loc;   // interface-face local nodes (sorted, < nc)
  };
  int maxlev = 0;
  for (int l : m->elem_level) maxlev = std::max(maxlev, l);
  std::vector<std::vector<IfaceElem>> inter(maxlev + 1);
  for (int iel = 0; iel < m->nel; iel++) {
    std::vector<int> loc;
    for (int f = 0; f < nf; f++)
      if (m->face_flag[(size_t)iel * nf + f] == -1 && seen[fslot[(size_t)iel * nf + f]] == 1)
        loc.insert(loc.end(), face_n[f], face_n[f] + face_nn[f]);
    if (loc.empty()) continue;
    std::sort(loc.begin(), loc.end());
    loc.erase(std::unique(loc.begin(), loc.end()), loc.end());
    inter[m->elem_level[iel]].push_back({iel, std::move(loc)});
  }
  const int ndof = mesh_ndofs(m, fe);
  std::vector<int> owner_level(ndof, -1);
  std::unordered_map<int, std::vector<std::pair<int, double>>> raw;
  std::map<int, std::map<int, double>> rest;       // reference mode: master -> {son: value}, ordered like the reference's std::map
  for (int Lc = 0; Lc <= maxlev; Lc++) {
    if (inter[Lc].empty()) continue;
    for (int Lf = Lc + 1; Lf <= maxlev; Lf++) {
      if (inter[Lf].empty()) continue;
      // fine interface nodes sorted by x for the box queries
      std::vector<int> ids;
      for (auto& ie : inter[Lf])
        for (int n : ie.loc) ids.push_back(m->elem_dof[(size_t)ie.iel * nl + n]);
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
      std::sort(ids.begin(), ids.end(), [&](int a, int b) {
        const double xa = m->coords[(size_t)a * dim], xb = m->coords[(size_t)b * dim];
        return xa < xb || (xa == xb && a < b);
      });
      std::vector<double> xs(ids.size());
      for (size_t k = 0; k < ids.size(); k++) xs[k] = m->coords[(size_t)ids[k] * dim];
      // (1) per coarse interface element, in parallel: the fine interface nodes inside it and the basis values there (box query, inverse map)
      struct Hit {
        int ldof;
        double phi[27];
      };
      const auto& cel = inter[Lc];
      std::vector<std::vector<Hit>> hits(cel.size());
      auto search = [&](size_t q0, size_t q1) {
        for (size_t q = q0; q < q1; q++) {
          const IfaceElem& ie = cel[q];
          const int* ed = &m->elem_dof[(size_t)ie.iel * nl];
          double xv[81], lo[3], hi[3];
          for (int d = 0; d < dim; d++) lo[d] = 1e300, hi[d] = -1e300;
          for (int i = 0; i < nl; i++)
            for (int d = 0; d < dim; d++) {
              const double c = m->coords[(size_t)ed[i] * dim + d];
              xv[i * dim + d] = c;
              lo[d] = std::min(lo[d], c);
              hi[d] = std::max(hi[d], c);
            }
          for (int d = 0; d < dim; d++) {
            const double pad = 0.01 * (hi[d] - lo[d]);
            lo[d] -= pad;
            hi[d] += pad;
          }
          const size_t k0 = std::lower_bound(xs.begin(), xs.end(), lo[0]) - xs.begin();
          for (size_t k = k0; k < ids.size() && xs[k] <= hi[0]; k++) {
            const int ldof = ids[k];
            const double* xp = &m->coords[(size_t)ldof * dim];
            bool in = true;
            for (int d = 1; d < dim; d++) in = in && xp[d] >= lo[d] && xp[d] <= hi[d];
            if (!in) continue;
            bool mine = false;
            for (int i = 0; i < nc; i++) mine = mine || ed[i] == ldof;
            if (mine) continue;
            double xi[3] = {0, 0, 0};
            if (!inverse_map_q2(geom, dim, xv, xp, xi)) continue;
            bool inside = true;
            for (int d = 0; d < dim; d++) inside = inside && std::fabs(xi[d]) <= 1.0 + 1e-4;
            if (!inside) continue;
            Hit h;
            h.ldof = ldof;
            eval_basis(geom, fe, xi, h.phi, nullptr);
            hits[q].push_back(h);
          }
        }
      };
      const int nth = cel.size() >= 256 ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
      if (nth == 1) {
        search(0, cel.size());
      } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; t++) th.emplace_back(search, cel.size() * t / nth, cel.size() * (t + 1) / nth);
        for (auto& x : th) x.join();
      }
      // (2) in the order of the coarse elements: which level describes a node (mode 1), the rows, the reference's map
      for (size_t q = 0; q < cel.size(); q++) {
        const IfaceElem& ie = cel[q];
        const int* ed = &m->elem_dof[(size_t)ie.iel * nl];
        for (const Hit& hit : hits[q]) {
          const int ldof = hit.ldof;
          const double* phi = hit.phi;
          if (m->amr_mode == 1) {
            if (owner_level[ldof] < 0) owner_level[ldof] = Lc;
            if (owner_level[ldof] != Lc) continue;
          }
          auto& row = raw[ldof];
          for (int n : ie.loc) {
            if (std::fabs(phi[n]) < 1.0e-10) continue;
            const int jd = ed[n];
            if (m->amr_mode == 0) {          // the reference's map restriction[master][son] with its diagonal marks (Mesh.cpp:1560-1567)
              auto& mrow = rest[jd];
              if (mrow.find(jd) == mrow.end()) mrow[jd] = 1.;
              mrow[ldof] = phi[n];
              rest[ldof][ldof] = 10.;
            }
            bool found = false;
            for (auto& e : row)
              if (e.first == jd) {
                e.second = phi[n];
                found = true;
              }
            if (!found) row.emplace_back(jd, phi[n]);
          }
        }
      }
    }
  }
  if (m->amr_mode == 0) {
    // second half of the reference function as written (Mesh.cpp:1711-1801): for every real master (diagonal mark < 5) a depth-first
    // walk through sons, grandsons, ...: restriction[master][son] += value * heredity(father); a son already present in the
    // genealogy lists of the levels above the one being filled is skipped ("alreadyFound").  For a node on the interfaces with two
    // coarser levels this keeps the direct entry and drops the path through the intermediate hanging node, so its row does not sum to
    // one -- that is the reference's result, reproduced here.
    const std::map<int, std::map<int, double>>& copy = rest;     // (read only from here on)
    std::map<int, std::vector<std::pair<int, double>>> hrow;        // hanging dof -> (master, weight)
    for (auto& kv : copy)
      if (kv.second.at(kv.first) > 5.) hrow[kv.first];
    std::vector<std::vector<int>> genealogy;
    std::vector<std::vector<double>> heredity;
    std::vector<size_t> index;
    for (auto& kv : copy) {
      const int inode = kv.first;
      if (!(kv.second.at(inode) < 5.)) continue;
      std::map<int, double> acc;
      genealogy.assign(1, std::vector<int>(1, inode));
      heredity.assign(1, std::vector<double>(1, 1.));
      index.assign(1, 0);
      size_t level = 1;
      while (level > 0) {
        const int father = genealogy[level - 1][index[level - 1]];
        const double hf = heredity[level - 1][index[level - 1]];
        genealogy.resize(level + 1);
        heredity.resize(level + 1);
        index.resize(level + 1);
        genealogy[level].clear();
        heredity[level].clear();
        index[level] = 0;
        for (auto& e : copy.at(father)) {
          const int son = e.first;
          bool found = false;
          for (size_t kl = 0; kl < level && !found; kl++)
            for (int g : genealogy[kl])
              if (g == son) {
                found = true;
                break;
              }
          if (found) continue;
          genealogy[level].push_back(son);
          heredity[level].push_back(e.second * hf);
          acc[son] += e.second * hf;
        }
        if (!genealogy[level].empty()) {
          level++;
        } else {
          bool test = true;
          while (test && level > 0) {
            index[level - 1]++;
            test = false;
            if (index[level - 1] == genealogy[level - 1].size()) {
              level--;
              test = true;
            }
          }
        }
      }
      for (auto& e : acc) hrow[e.first].emplace_back(inode, e.second);
    }
    for (auto& kv : hrow) {
      out.hang.push_back(kv.first);
      std::sort(kv.second.begin(), kv.second.end());
      for (auto& e : kv.second) {
        out.master.push_back(e.first);
        out.w.push_back(e.second);
      }
      out.ptr.push_back((int)out.master.size());
    }
    return;
  }
  // resolve masters that hang themselves (depth-first, masters in increasing dof order)
  std::unordered_map<int, std::vector<std::pair<int, double>>> res;
  std::function<const std::vector<std::pair<int, double>>&(int, int)> expand = [&](int l, int depth) -> const std::vector<std::pair<int, double>>& {
    auto it = res.find(l);
    if (it != res.end()) return it->second;
    std::vector<std::pair<int, double>> row = raw[l];
    std::sort(row.begin(), row.end());
    std::vector<std::pair<int, double>> acc;
    auto add = [&](int j, double w) {
      for (auto& e : acc)
        if (e.first == j) {
          e.second += w;
          return;
        }
      acc.emplace_back(j, w);
    };
    for (auto& e : row) {
      if (raw.count(e.first) && depth < 16) {
        const auto sub = expand(e.first, depth + 1);   // copy: the map may rehash below
        for (auto& s : sub) add(s.first, e.second * s.second);
      } else {
        add(e.first, e.second);
      }
    }
    std::sort(acc.begin(), acc.end());
    return res.emplace(l, std::move(acc)).first->second;
  };
  std::vector<int> hang;
  for (auto& kv : raw) hang.push_back(kv.first);
  std::sort(hang.begin(), hang.end());
  for (int l : hang) {
    const auto& row = expand(l, 0);
    out.hang.push_back(l);
    for (auto& e : row) {
      out.master.push_back(e.first);
      out.w.push_back(e.second);
    }
    out.ptr.push_back((int)out.master.size());
  }
}

extern "C" int fh_mesh_set_amr_mode(fh_mesh_t m, int mode) {
  FH_REQUIRE(m && (mode == 0 || mode == 1), "fh_mesh_set_amr_mode: mode must be 0 (as the reference computes it) or 1 (coarsest level, rows sum to one)");
  m->amr_mode = mode;
  m->amr_cache[0].reset();
  m->amr_cache[1].reset();
  return 0;
}

extern "C" int fh_mesh_amr_constraints(fh_mesh_t m, int fe, int* n_hanging, int* nnz, int* hanging, int* ptr, int* master, double* weight) {
  FH_GUARD_BEGIN
  FH_REQUIRE(m && n_hanging && nnz, "fh_mesh_amr_constraints: null argument");
  FH_REQUIRE(fe == 0 || fe == 2, "fh_mesh_amr_constraints: fe must be 0 or 2");
  AmrRows R;
  amr_constraints(m, fe, R);
  if (hanging) {
    FH_REQUIRE(*n_hanging >= (int)R.hang.size() && *nnz >= (int)R.master.size(), "fh_mesh_amr_constraints: capacity too small");
    fh_copy_out(hanging, R.hang);
    if (ptr) fh_copy_out(ptr, R.ptr);
    if (master) fh_copy_out(master, R.master);
    if (weight) fh_copy_out(weight, R.w);
  }
  *n_hanging = (int)R.hang.size();
  *nnz = (int)R.master.size();
  return 0;
  FH_GUARD_END("fh_mesh_amr_constraints")
}

// P_amr (n x n): identity rows for regular dofs; a hanging dof's row holds its master weights and an explicit zero on
// the diagonal (the reference inserts restriction[son][son] = 0, which keeps (son, son) in the pattern of P^T K P so
// that SetPenalty can put its 1 there)
extern "C" int fh_build_amr_prolongator(fh_ctx_t ctx, fh_mesh_t m, int fe, fh_mat_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && m && out, "fh_build_amr_prolongator: null argument");
  FH_REQUIRE(fe == 0 || fe == 2, "fh_build_amr_prolongator: fe must be 0 or 2");
  AmrRows R;
  amr_constraints(m, fe, R);
  const int n = mesh_ndofs(m, fe);
  std::vector<int> rowptr(n + 1, 0), col;
  std::vector<double> val;
  size_t h = 0;
  std::vector<std::pair<int, double>> row;
  for (int i = 0; i < n; i++) {
    if (h < R.hang.size() && R.hang[h] == i) {
      row.clear();
      row.emplace_back(i, 0.0);
      for (int k = R.ptr[h]; k < R.ptr[h + 1]; k++) row.emplace_back(R.master[k], R.w[k]);
      std::sort(row.begin(), row.end());
      for (auto& e : row) {
        col.push_back(e.first);
        val.push_back(e.second);
      }
      h++;
    } else {
      col.push_back(i);
      val.push_back(1.0);
    }
    rowptr[i + 1] = (int)col.size();
  }
  return fh_mat_create_csr(ctx, n, n, rowptr.data(), col.data(), val.data(), out);
  FH_GUARD_END("fh_build_amr_prolongator")
}

// ---------------------------------------------------------------------------------------------------------------------
// multi-variable systems (a9, a21): LinearEquation::GetSystemDof (LinearEquation.cpp:76-85) with the nprocs = 1 offsets
// KKoffset[k] = sum of the sizes of the variables before k (:212-237).  Variables are Lagrange families 0 (Q1) or 2 (Q2).
// ---------------------------------------------------------------------------------------------------------------------
// fe: 0 LAGRANGE FIRST, 2 LAGRANGE SECOND; with_pw also 4 = DISCONTINUOUS_POLYNOMIAL FIRST (1, xi, eta (, zeta) on the reference element, dim + 1 dofs owned
// by every element: Mesh::GetSolutionDof for solution type 4 on one process = i * nel + iel)
static int check_vars(const char* who, int nvars, const int* fe, bool with_pw = false) {
  FH_REQUIRE(nvars >= 1 && nvars <= 8 && fe, "%s: 1..8 variables expected", who);
  for (int k = 0; k < nvars; k++)
    FH_REQUIRE(fe_known(fe[k]) || (with_pw && fe[k] == 4), "%s: variable %d: fe must be 0 .. 3%s", who, k, with_pw ? " (or 4, discontinuous linear)" : "");
  return 0;
}
static int var_elem_dofs(const fh_mesh_s* m, int fe) { return fe == 4 ? m->dim + 1 : ndofs_of(m->geom, fe); }
static int var_mesh_dofs(const fh_mesh_s* m, int fe) { return fe == 4 ? (m->dim + 1) * m->nel : mesh_ndofs(m, fe); }

extern "C" int fh_system_elem_dofs(fh_mesh_t m, int nvars, const int* fe, int* nd_out, int* offsets, int* elem_sys) {
  FH_GUARD_BEGIN
  FH_REQUIRE(m, "fh_system_elem_dofs: null mesh");
  FH_TRY(check_vars("fh_system_elem_dofs", nvars, fe, true));
  int nd = 0;
  int64_t off = 0;
  std::vector<int> offs(nvars + 1, 0);
  for (int k = 0; k < nvars; k++) {
    nd += var_elem_dofs(m, fe[k]);
    off += var_mesh_dofs(m, fe[k]);
    FH_REQUIRE(off < 2147483647ll, "fh_system_elem_dofs: the system does not fit 32-bit ids");
    offs[k + 1] = (int)off;
  }
  if (nd_out) *nd_out = nd;
  if (offsets) fh_copy_out(offsets, offs);
  if (elem_sys)
    for (int iel = 0; iel < m->nel; iel++) {
      int p = 0;
      for (int k = 0; k < nvars; k++)
        for (int i = 0; i < var_elem_dofs(m, fe[k]); i++)
          elem_sys[(size_t)iel * nd + p++] = offs[k] + (fe[k] == 4 ? i * m->nel + iel : fe[k] == FE_CONSTANT ? iel : m->elem_dof[(size_t)iel * m->nloc + i]);
    }
  return 0;
  FH_GUARD_END("fh_system_elem_dofs")
}

// BuildProlongatorMatrix loops the system variables (LinearImplicitSystem.cpp:890-905): block-diagonal interpolation,
// one block per variable; Dirichlet rows / columns are zeroed by the caller (fh_mat_zero_rows / fh_mat_zero_cols)
extern "C" int fh_build_system_prolongator(fh_ctx_t ctx, fh_mesh_t mc, fh_mesh_t mf, int nvars, const int* fe, fh_mat_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && mc && mf && out, "fh_build_system_prolongator: null argument");
  FH_TRY(check_vars("fh_build_system_prolongator", nvars, fe, true));
  fh_mat_t blk[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < nvars; k++)
    if (fe[k] != 4 && !blk[fe[k]]) FH_TRY(fh_build_prolongator(ctx, mc, mf, fe[k], 0, &blk[fe[k]]));
  // the block of a discontinuous linear variable: the element prolongator of solution type 4 (ElemType.cpp:446-520) -- the coarse function at the centre of
  // the child for its constant, half the coarse slope for its linear functions; an element carried over unrefined keeps its three / four functions
  std::vector<int> pw_rp, pw_col;
  std::vector<double> pw_val;
  const int npw = mc->dim + 1, nch = nvert_of(mc->geom);
  bool any_pw = false;
  for (int k = 0; k < nvars; k++) any_pw |= fe[k] == 4;
  if (any_pw) {
    FH_REQUIRE((int)mc->child.size() == mc->nel * nch, "fh_build_system_prolongator: the coarse mesh has not been refined into the fine one");
    std::vector<int> parent(mf->nel, -1), which(mf->nel, -1);
    for (int e = 0; e < mc->nel; e++)
      for (int j = 0; j < nch; j++) {
        const int jel = mc->child[(size_t)e * nch + j];
        if (jel < 0) continue;
        FH_REQUIRE(jel < mf->nel, "fh_build_system_prolongator: child element %d out of range", jel);
        parent[jel] = e;
        which[jel] = mc->refined[e] ? j : -1;
      }
    for (int j = 0; j < nch; j++) FH_REQUIRE(fine2coarse_vertex(mc->geom, j, j) == j, "fh_build_system_prolongator: child %d does not sit at vertex %d of its father", j, j);
    pw_rp.assign((size_t)npw * mf->nel + 1, 0);
    for (int f = 0; f < npw; f++)
      for (int jel = 0; jel < mf->nel; jel++) {
        FH_REQUIRE(parent[jel] >= 0, "fh_build_system_prolongator: fine element %d has no father", jel);
        const int e = parent[jel], j = which[jel];
        if (f == 0 && j >= 0) {
          pw_col.push_back(e);
          pw_val.push_back(1.0);
          for (int d = 0; d < mc->dim; d++) {
            pw_col.push_back((1 + d) * mc->nel + e);
            pw_val.push_back(0.5 * xc(mc->geom, j, d));      // the child's centre in the reference coordinates of its father
          }
        } else {
          pw_col.push_back(f * mc->nel + e);
          pw_val.push_back(j >= 0 ? 0.5 : 1.0);
        }
        pw_rp[(size_t)f * mf->nel + jel + 1] = (int)pw_col.size();
      }
  }
  int nf = 0, ncc = 0;
  int64_t nnz = 0;
  for (int k = 0; k < nvars; k++) {
    nf += fe[k] == 4 ? npw * mf->nel : blk[fe[k]]->m;
    ncc += fe[k] == 4 ? npw * mc->nel : blk[fe[k]]->n;
    nnz += fe[k] == 4 ? (int64_t)pw_col.size() : blk[fe[k]]->nnz;
  }
  FH_REQUIRE(nnz < 2147483647ll, "fh_build_system_prolongator: nnz overflows int32");
  std::vector<int> rowptr(nf + 1, 0), col((size_t)nnz);
  std::vector<double> val((size_t)nnz);
  int r0 = 0, c0 = 0, p = 0;
  for (int k = 0; k < nvars; k++) {
    if (fe[k] == 4) {
      const int m4 = npw * mf->nel;
      for (int i = 0; i < m4; i++) {
        for (int q = pw_rp[i]; q < pw_rp[i + 1]; q++) {
          col[p] = c0 + pw_col[q];
          val[p++] = pw_val[q];
        }
        rowptr[r0 + i + 1] = p;
      }
      r0 += m4;
      c0 += npw * mc->nel;
      continue;
    }
    fh_mat_t B = blk[fe[k]];
    std::vector<double> bv(B->nnz);
    FH_CHECK_HIP(hipMemcpy(bv.data(), B->d_val, (size_t)B->nnz * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < B->m; i++) {
      for (int q = B->h_rowptr[i]; q < B->h_rowptr[i + 1]; q++) {
        col[p] = c0 + fh_hcol(B)[q];
        val[p++] = bv[q];
      }
      rowptr[r0 + i + 1] = p;
    }
    r0 += B->m;
    c0 += B->n;
  }
  for (auto b : blk)
    if (b) fh_mat_destroy(b);
  return fh_mat_create_csr(ctx, nf, ncc, rowptr.data(), col.data(), val.data(), out);
  FH_GUARD_END("fh_build_system_prolongator")
}

// Blocks of the Schwarz (Vanka) smoother for saddle-point systems, the GPU form of the element-block ASM of
// petsc_asm/LinearEquationSolverPetscAsm.cpp:91-276 with one block per dof of the LAST variable (the "Schur" variable,
// :223-256): that dof plus, for every other variable, all dofs of the elements that own it (:134-170).  Two-call
// protocol: ptr == NULL returns the counts.
extern "C" int fh_mesh_vertex_patches(fh_mesh_t m, int nvars, const int* fe, int* npatch, int* total, int* ptr, int* dofs) {
  FH_GUARD_BEGIN
  FH_REQUIRE(m && npatch && total, "fh_mesh_vertex_patches: null argument");
  FH_TRY(check_vars("fh_mesh_vertex_patches", nvars, fe));
  FH_REQUIRE(nvars >= 2, "fh_mesh_vertex_patches: needs at least one non-Schur variable and the Schur variable");
  const int nl = m->nloc, ns = mesh_ndofs(m, fe[nvars - 1]), ncs = ndofs_of(m->geom, fe[nvars - 1]);
  std::vector<int> offs(nvars + 1, 0);
  for (int k = 0; k < nvars; k++) offs[k + 1] = offs[k] + mesh_ndofs(m, fe[k]);
  // elements of every Schur dof
  std::vector<int> eptr(ns + 1, 0);
  for (int iel = 0; iel < m->nel; iel++)
    for (int i = 0; i < ncs; i++) eptr[m->elem_dof[(size_t)iel * nl + i] + 1]++;
  for (int v = 0; v < ns; v++) eptr[v + 1] += eptr[v];
  std::vector<int> eadj(eptr[ns]), cur(eptr.begin(), eptr.end() - 1);
  for (int iel = 0; iel < m->nel; iel++)
    for (int i = 0; i < ncs; i++) eadj[cur[m->elem_dof[(size_t)iel * nl + i]]++] = iel;
  std::vector<int> out_ptr(1, 0), out_dofs, nodes;
  for (int v = 0; v < ns; v++) {
    for (int k = 0; k < nvars - 1; k++) {
      const int nck = ndofs_of(m->geom, fe[k]);
      nodes.clear();
      for (int q = eptr[v]; q < eptr[v + 1]; q++)
        for (int i = 0; i < nck; i++) nodes.push_back(m->elem_dof[(size_t)eadj[q] * nl + i]);
      std::sort(nodes.begin(), nodes.end());
      nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
      for (int nd : nodes) out_dofs.push_back(offs[k] + nd);
    }
    out_dofs.push_back(offs[nvars - 1] + v);
    out_ptr.push_back((int)out_dofs.size());
  }
  if (ptr) {
    FH_REQUIRE(*npatch >= ns && *total >= (int)out_dofs.size(), "fh_mesh_vertex_patches: capacity too small");
    fh_copy_out(ptr, out_ptr);
    if (dofs) fh_copy_out(dofs, out_dofs);
  }
  *npatch = ns;
  *total = (int)out_dofs.size();
  return 0;
  FH_GUARD_END("fh_mesh_vertex_patches")
}

// ---------------------------------------------------------------------------------------------------------------------
// Gambit neutral files: GambitIO::read
//   src/06_mesh/00_single_level/01_input/01_from_external_file/GambitIO.cpp:93-352
// for the element types of this path (HEX27, QUAD9; every node of such an element is in the file, so
// AddBiquadraticNodesNotInMeshFile, Mesh.cpp:1207-1333, has nothing to add).  The token-stream reading follows the
// reference: control data after "NDFVL", elements after "ELEMENTS/CELLS", coordinates after "COORDINATES", groups after
// each "GROUP:" (group and material numbers are integers), boundary sets after each "CONDITIONS" (set number n gives the
// face flag -n-1, GambitIO.cpp:337).  Gambit's local node order is lexicographic (xi slowest, eta, zeta fastest and
// descending) for the 27-node brick and perimeter-then-centre for the 9-node quadrilateral; its brick faces 1..6 are FEMuS
// faces 0,4,2,5,3,1 (GambitIO.cpp:55-89).  After reading: elements sorted by (material, group, file order) as
// Mesh::mesh_reorder_elem_quantities does (Mesh.cpp:626-690), then the first-touch node numbering (Mesh.cpp:517-559).
// ---------------------------------------------------------------------------------------------------------------------
#include <fstream>

static int read_gambit(const char* path, double Lref, fh_mesh_t* out);

// the file is untrusted input: sizes are checked before they size anything, the mesh object is owned by a guard until it is handed
// out, and no C++ exception (bad_alloc, length_error) crosses the C boundary
extern "C" int fh_mesh_read_gambit(const char* path, double Lref, fh_mesh_t* out) {
  FH_REQUIRE(path && out && Lref != 0.0, "fh_mesh_read_gambit: bad arguments");
  try {
    return read_gambit(path, Lref, out);
  } catch (const std::exception& e) {
    fh_set_error("fh_mesh_read_gambit: %s: %s", path, e.what());
    return 2;
  }
}

static int read_gambit(const char* path, double Lref, fh_mesh_t* out) {
  std::ifstream inf(path);
  FH_REQUIRE((bool)inf, "Generic-mesh file %s can not read parameters", path);
  std::vector<std::string> tok;
  {
    std::string t;
    while (inf >> t) tok.push_back(t);
  }
  size_t p = 0;
  auto seek = [&](const char* word, size_t from) {
    for (size_t k = from; k < tok.size(); k++)
      if (tok[k] == word) return k;
    return tok.size();
  };
  auto num = [&](size_t k, double* v) {
    if (k >= tok.size()) return false;
    char* e = nullptr;
    *v = strtod(tok[k].c_str(), &e);
    return e != tok[k].c_str() && *e == 0;
  };
  // a token used as an integer: finite, integral, inside [lo, hi] -- (int) of NaN or of 1e300 is undefined behaviour, not an error path
  auto whole = [&](size_t k, double lo, double hi, double* v) { return num(k, v) && *v >= lo && *v <= hi && *v == std::floor(*v); };
  p = seek("NDFVL", 0);
  double v[6];
  for (int k = 0; k < 6; k++) FH_REQUIRE(num(p + 1 + k, &v[k]), "fh_mesh_read_gambit: %s: error control data mesh", path);
  FH_REQUIRE(p + 7 < tok.size() && tok[p + 7] == "ENDOFSECTION", "fh_mesh_read_gambit: %s: error control data mesh", path);
  // every count is bounded by the number of tokens in the file (a node, an element, a group, a boundary set each take tokens)
  for (int k = 0; k < 6; k++)
    FH_REQUIRE(v[k] >= 0.0 && v[k] <= (double)tok.size() && v[k] == std::floor(v[k]), "fh_mesh_read_gambit: %s: error control data mesh (count %g)", path, v[k]);
  const int nvt = (int)v[0], nel = (int)v[1], ngroup = (int)v[2], nbcd = (int)v[3], dim = (int)v[4], dimNodes = (int)v[5];
  FH_REQUIRE(dim == 2 || dim == 3, "fh_mesh_read_gambit: %s: %d-dimensional meshes are not served (HEX27 / QUAD9 only)", path, dim);
  FH_REQUIRE(nvt >= 1 && nel >= 1 && dimNodes >= dim && dimNodes <= 3, "fh_mesh_read_gambit: %s: error control data mesh", path);
  std::unique_ptr<fh_mesh_s> guard(new fh_mesh_s());
  fh_mesh_s* m = guard.get();
  m->geom = dim == 3 ? GEOM_HEX : GEOM_QUAD;
  m->dim = dim;
  m->nloc = nloc_of(m->geom);
  m->nel = nel;
  const int nl = m->nloc, nf = nfaces_of(m->geom);
  // Gambit local node -> FEMuS local node
  std::vector<int> g2f(nl);
  if (dim == 3) {
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        for (int c = 0; c < 3; c++)
          for (int n = 0; n < nl; n++)
            if (xc(m->geom, n, 0) == a - 1 && xc(m->geom, n, 1) == b - 1 && xc(m->geom, n, 2) == 1 - c) g2f[a * 9 + b * 3 + c] = n;
  } else {
    for (int g = 0; g < 8; g++) g2f[g] = (g % 2 == 0) ? g / 2 : 4 + g / 2;
    g2f[8] = 8;
  }
  const int gface_hex[6] = {0, 4, 2, 5, 3, 1};
  // elements
  p = seek("ELEMENTS/CELLS", 0);
  FH_REQUIRE(p < tok.size(), "fh_mesh_read_gambit: %s: no ELEMENTS/CELLS section", path);
  p += 2;
  std::vector<int> ed((size_t)nel * nl);
  for (int iel = 0; iel < nel; iel++) {
    double nve;
    FH_REQUIRE(whole(p + 2, 0, 1000, &nve), "fh_mesh_read_gambit: %s: error element data mesh", path);
    if ((int)nve != nl) {
      fh_set_error("Error! Invalid element type in reading Gambit File! (element %d has %d nodes; HEX27 / QUAD9 meshes are served)", iel + 1, (int)nve);
      return 2;
    }
    p += 3;
    for (int i = 0; i < nl; i++) {
      double val;
      FH_REQUIRE(whole(p + i, 1, nvt, &val), "fh_mesh_read_gambit: %s: bad node id in element %d", path, iel + 1);
      ed[(size_t)iel * nl + g2f[i]] = (int)val - 1;
    }
    p += nl;
  }
  FH_REQUIRE(p < tok.size() && tok[p] == "ENDOFSECTION", "fh_mesh_read_gambit: %s: error element data mesh", path);
  // coordinates
  p = seek("COORDINATES", 0);
  FH_REQUIRE(p < tok.size(), "fh_mesh_read_gambit: %s: no NODAL COORDINATES section", path);
  p += 2;
  std::vector<double> xyz((size_t)nvt * dim);
  for (int j = 0; j < nvt; j++) {
    for (int d = 0; d < dimNodes; d++) {
      double c;
      FH_REQUIRE(num(p + 1 + d, &c) && std::isfinite(c), "fh_mesh_read_gambit: %s: error node data mesh", path);
      if (d < dim) xyz[(size_t)j * dim + d] = c / Lref;
    }
    p += 1 + dimNodes;
  }
  FH_REQUIRE(p < tok.size() && tok[p] == "ENDOFSECTION", "fh_mesh_read_gambit: %s: error node data mesh 1", path);
  // groups and materials
  std::vector<int> group(nel, 1), material(nel, 0);
  p = 0;
  for (int k = 0; k < ngroup; k++) {
    p = seek("GROUP:", p);
    double ngel, mat, name;
    FH_REQUIRE(p < tok.size() && whole(p + 3, 0, nel, &ngel) && whole(p + 5, -1e9, 1e9, &mat) && whole(p + 8, -1e9, 1e9, &name),
               "fh_mesh_read_gambit: %s: error group data mesh", path);
    p += 10;
    for (int i = 0; i < (int)ngel; i++) {
      double iel;
      FH_REQUIRE(whole(p + i, 1, nel, &iel), "fh_mesh_read_gambit: %s: error group data mesh", path);
      group[(int)iel - 1] = (int)name;
      material[(int)iel - 1] = (int)mat;
    }
    p += (size_t)ngel;
    FH_REQUIRE(p < tok.size() && tok[p] == "ENDOFSECTION", "fh_mesh_read_gambit: %s: error group data mesh", path);
  }
  // boundary sets
  std::vector<int> ff((size_t)nel * nf, -1);
  p = 0;
  for (int k = 0; k < nbcd; k++) {
    p = seek("CONDITIONS", p);
    double value, nface;
    FH_REQUIRE(p < tok.size() && whole(p + 2, -1e9, 1e9, &value) && whole(p + 4, 0, (double)nel * nf, &nface),
               "fh_mesh_read_gambit: %s: error boundary data mesh", path);
    p += 7;
    for (int i = 0; i < (int)nface; i++) {
      double iel, iface;
      FH_REQUIRE(whole(p, 1, nel, &iel) && whole(p + 2, 1, nf, &iface), "fh_mesh_read_gambit: %s: error boundary data mesh", path);
      const int f = dim == 3 ? gface_hex[(int)iface - 1] : (int)iface - 1;
      ff[((size_t)iel - 1) * nf + f] = -(int)value - 1;
      p += 3;
    }
    FH_REQUIRE(p < tok.size() && tok[p] == "ENDOFSECTION", "fh_mesh_read_gambit: %s: error boundary data mesh", path);
  }
  // element order: (material, group, file order)
  std::vector<int> order(nel);
  for (int i = 0; i < nel; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (material[a] != material[b]) return material[a] < material[b];
    return group[a] < group[b];
  });
  m->elem_dof.resize((size_t)nel * nl);
  m->face_flag.resize((size_t)nel * nf);
  for (int i = 0; i < nel; i++) {
    memcpy(&m->elem_dof[(size_t)i * nl], &ed[(size_t)order[i] * nl], nl * sizeof(int));
    memcpy(&m->face_flag[(size_t)i * nf], &ff[(size_t)order[i] * nf], nf * sizeof(int));
  }
  m->coords = xyz;
  m->elem_level.assign(nel, 0);
  m->elem_group.resize(nel);
  m->elem_material.resize(nel);
  for (int i = 0; i < nel; i++) {
    m->elem_group[i] = group[order[i]];
    m->elem_material[i] = material[order[i]];
  }
  {
    std::vector<char> used(nvt, 0);
    for (int x : m->elem_dof) used[x] = 1;
    for (int j = 0; j < nvt; j++)
      if (!used[j]) {
        fh_set_error("fh_mesh_read_gambit: %s: node %d belongs to no element", path, j + 1);
        return 2;
      }
  }
  first_touch_renumber(*m, nvt);
  *out = guard.release();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Domain decomposition of ARBITRARY coarse meshes (the reference: METIS_PartMeshDual on the coarsest level, ncommon = dim + 1 over the
// element nodes, i.e. face neighbours; children inherit the parent's rank, MeshMetisPartitioning.cpp:71-113, 143-155).  METIS is not
// available (and seeds its partitions randomly), so the partition is native: recursive bisection of the dual graph by breadth-first
// growing from a pseudo-peripheral element -- balanced to one element, connected where the graph allows it, deterministic.
//   fh_mesh_partition      part[nel] of the coarse mesh
//   fh_mesh_rank_elements  a rank's elements: the ones it owns, then the ring of elements sharing a node with them (ascending)
//   fh_mesh_submesh        a FEMuS-numbered mesh of a list of elements (first-touch renumbering, Mesh.cpp:517-559) + the node map
//   fh_dd_topo_node_keys   global id and owner of every node of a refined level of such a sub-mesh, WITHOUT coordinates: a node lies
//                          inside exactly one entity of the coarse mesh (vertex, edge, face, element = one of its 27 / 9 nodes), at
//                          dyadic offsets measured in a frame fixed by the global ids of the entity's corners; owner = the lowest
//                          rank among the coarse elements sharing the entity (Mesh.cpp:517-559)
// ------------------------------------------------------------------------------------------------------------------
static void dual_graph(const fh_mesh_s* G, std::vector<int>& ptr, std::vector<int>& adj) {
  // two elements are face neighbours iff they share the node in the middle of a face (HEX27: local nodes 20..25, QUAD9: 4..7)
  const int nl = G->nloc, f0 = G->geom == GEOM_HEX ? 20 : 4, f1 = G->geom == GEOM_HEX ? 26 : 8;
  std::unordered_map<int, int> first;
  std::vector<std::pair<int, int> > edges;
  for (int e = 0; e < G->nel; e++)
    for (int i = f0; i < f1; i++) {
      const int nd = G->elem_dof[(size_t)e * nl + i];
      auto it = first.find(nd);
      if (it == first.end()) first[nd] = e;
      else {
        edges.emplace_back(it->second, e);
        edges.emplace_back(e, it->second);
      }
    }
  std::sort(edges.begin(), edges.end());
  edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
  ptr.assign(G->nel + 1, 0);
  for (auto& ed : edges) ptr[ed.first + 1]++;
  for (int e = 0; e < G->nel; e++) ptr[e + 1] += ptr[e];
  adj.resize(edges.size());
  for (size_t k = 0; k < edges.size(); k++) adj[k] = edges[k].second;
}

static void bfs_order(const std::vector<int>& ptr, const std::vector<int>& adj, const std::vector<int>& set, const std::vector<char>& in, int start,
                      std::vector<int>& order) {
  std::vector<char> seen(in.size(), 0);
  order.clear();
  size_t next_seed = 0;
  int seed = start;
  while (order.size() < set.size()) {
    if (seed < 0) {
      while (next_seed < set.size() && seen[set[next_seed]]) next_seed++;
      seed = set[next_seed];
    }
    size_t head = order.size();
    order.push_back(seed);
    seen[seed] = 1;
    while (head < order.size()) {
      const int e = order[head++];
      for (int k = ptr[e]; k < ptr[e + 1]; k++) {
        const int f = adj[k];
        if (in[f] && !seen[f]) {
          seen[f] = 1;
          order.push_back(f);
        }
      }
    }
    seed = -1;
  }
}

// where the prefix of an ordering is cut: np1 / np of the elements, or -- with weights (the work below an element after adaptive
// refinement) -- the prefix whose weight is closest to np1 / np of the total; every side keeps at least as many elements as it has parts
static size_t split_point(const std::vector<int>& order, int np, int np1, const double* w) {
  size_t n1 = (order.size() * (size_t)np1 + np / 2) / np;
  if (w) {
    double total = 0.0;
    for (int e : order) total += w[e];
    const double want = total * (double)np1 / (double)np;
    double acc = 0.0;
    n1 = 0;
    while (n1 < order.size() && acc + 0.5 * w[order[n1]] <= want) acc += w[order[n1++]];
    const size_t lo = std::min(order.size(), (size_t)np1), hi = order.size() > (size_t)(np - np1) ? order.size() - (size_t)(np - np1) : 0;
    n1 = std::max(lo, std::min(n1, std::max(lo, hi)));
  }
  return n1;
}

// faces of the dual graph cut by taking the first n1 elements of an ordering of `set`
static long cut_of(const std::vector<int>& ptr, const std::vector<int>& adj, const std::vector<int>& order, size_t n1, const std::vector<char>& in,
                   std::vector<char>& side) {
  for (size_t k = 0; k < order.size(); k++) side[order[k]] = k < n1 ? 1 : 2;
  long cut = 0;
  for (size_t k = 0; k < n1; k++) {
    const int e = order[k];
    for (int q = ptr[e]; q < ptr[e + 1]; q++)
      if (in[adj[q]] && side[adj[q]] == 2) cut++;
  }
  return cut;
}

// Two orderings compete at every bisection and the one that cuts fewer faces wins (ties: the first):
//   (a) breadth-first growing from a pseudo-peripheral element -- needs nothing but the dual graph, follows bent and branched domains;
//   (b) the elements sorted along the principal axis of their (weighted) centroids (inertial bisection) -- a plane across the short way
//       of a compact block, where the level sets of (a) run diagonally and cut about twice as many faces.
struct PartGeom { const double* xc; int dim; };       // element centroids [nel * dim], or null

static void bisect(const std::vector<int>& ptr, const std::vector<int>& adj, std::vector<int>& set, int p0, int np, std::vector<int>& part,
                   const double* w = nullptr, const PartGeom* geo = nullptr) {
  if (np == 1 || set.empty()) {
    for (int e : set) part[e] = p0;
    return;
  }
  std::vector<char> in(part.size(), 0);
  for (int e : set) in[e] = 1;
  std::vector<int> order;
  bfs_order(ptr, adj, set, in, set[0], order);
  const int far1 = order.back();                       // pseudo-peripheral element: the far end of a breadth-first search, twice
  bfs_order(ptr, adj, set, in, far1, order);
  const int far2 = order.back();
  bfs_order(ptr, adj, set, in, far2, order);
  const int np1 = np / 2;
  size_t n1 = split_point(order, np, np1, w);
  if (geo && geo->xc && set.size() > 2) {
    const int dim = geo->dim;
    double mean[3] = {0, 0, 0}, wsum = 0.0;
    for (int e : set) {
      const double we = w ? w[e] : 1.0;
      for (int d = 0; d < dim; d++) mean[d] += we * geo->xc[(size_t)e * dim + d];
      wsum += we;
    }
    for (int d = 0; d < dim; d++) mean[d] /= wsum;
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int e : set) {
      const double we = w ? w[e] : 1.0;
      double x[3] = {0, 0, 0};
      for (int d = 0; d < dim; d++) x[d] = geo->xc[(size_t)e * dim + d] - mean[d];
      for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) C[i][j] += we * x[i] * x[j];
    }
    // principal axis by power iteration from the diagonal's largest direction (deterministic); a cube has no preferred axis: any is a plane cut
    double v[3] = {0, 0, 0};
    int dmax = 0;
    for (int d = 1; d < dim; d++)
      if (C[d][d] > C[dmax][dmax] * (1.0 + 1e-9)) dmax = d;
    v[dmax] = 1.0;
    for (int it = 0; it < 60; it++) {
      double u[3] = {0, 0, 0}, nrm = 0.0;
      for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) u[i] += C[i][j] * v[j];
      for (int i = 0; i < dim; i++) nrm += u[i] * u[i];
      nrm = sqrt(nrm);
      if (!(nrm > 0.0)) break;
      for (int i = 0; i < dim; i++) v[i] = u[i] / nrm;
    }
    std::vector<std::pair<double, int> > key(set.size());
    for (size_t k = 0; k < set.size(); k++) {
      const int e = set[k];
      double t = 0.0;
      for (int d = 0; d < dim; d++) t += v[d] * (geo->xc[(size_t)e * dim + d] - mean[d]);
      key[k] = std::make_pair(t, e);
    }
    // elements of one layer across the axis have equal projections up to rounding: quantised, so that the order inside a layer is the
    // element order and the cut is the same on every machine
    double span = 0.0;
    for (auto& kv : key) span = std::max(span, fabs(kv.first));
    const double q = span > 0.0 ? span * 1e-9 : 1.0;
    for (auto& kv : key) kv.first = std::floor(kv.first / q + 0.5);
    std::sort(key.begin(), key.end());
    std::vector<int> order2(set.size());
    for (size_t k = 0; k < key.size(); k++) order2[k] = key[k].second;
    const size_t n2 = split_point(order2, np, np1, w);
    std::vector<char> side(part.size(), 0);
    const long cut1 = cut_of(ptr, adj, order, n1, in, side), cut2 = cut_of(ptr, adj, order2, n2, in, side);
    if (cut2 < cut1) {
      order.swap(order2);
      n1 = n2;
    }
  }
  std::vector<int> a(order.begin(), order.begin() + n1), b(order.begin() + n1, order.end());
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  bisect(ptr, adj, a, p0, np1, part, w, geo);
  bisect(ptr, adj, b, p0 + np1, np - np1, part, w, geo);
}

static void elem_centroids_of(const fh_mesh_s* G, std::vector<double>& xc) {
  const int nv = nvert_of(G->geom), dim = G->dim;
  xc.assign((size_t)G->nel * dim, 0.0);
  for (int e = 0; e < G->nel; e++)
    for (int i = 0; i < nv; i++) {
      const int nd = G->elem_dof[(size_t)e * G->nloc + i];
      for (int d = 0; d < dim; d++) xc[(size_t)e * dim + d] += G->coords[(size_t)nd * dim + d] / nv;
    }
}

extern "C" int fh_mesh_partition(fh_mesh_t G, int nparts, int* part) {
  FH_GUARD_BEGIN
  FH_REQUIRE(G && part && nparts >= 1, "fh_mesh_partition: bad arguments");
  std::vector<int> ptr, adj, set(G->nel), p(G->nel, 0);
  dual_graph(G, ptr, adj);
  for (int e = 0; e < G->nel; e++) set[e] = e;
  std::vector<double> xc;
  elem_centroids_of(G, xc);
  const PartGeom geo{xc.data(), G->dim};
  bisect(ptr, adj, set, 0, nparts, p, nullptr, &geo);
  fh_copy_out(part, p);
  return 0;
  FH_GUARD_END("fh_mesh_partition")
}

// the same bisection with element weights: the parts balance the SUM of the weights instead of the element count -- what the reference
// gets from re-partitioning an adaptively refined level (MeshMetisPartitioning.cpp:41-113 with AMR = true) while children still inherit
// their coarse element's rank (:143-155): weight = number of finest-level descendants of the coarse element
extern "C" int fh_mesh_partition_weighted(fh_mesh_t G, int nparts, const double* weight, int* part) {
  FH_GUARD_BEGIN
  FH_REQUIRE(G && part && weight && nparts >= 1, "fh_mesh_partition_weighted: bad arguments");
  for (int e = 0; e < G->nel; e++) FH_REQUIRE(weight[e] > 0.0 && weight[e] < 1e300, "fh_mesh_partition_weighted: weight of element %d is not positive and finite", e);
  std::vector<int> ptr, adj, set(G->nel), p(G->nel, 0);
  dual_graph(G, ptr, adj);
  for (int e = 0; e < G->nel; e++) set[e] = e;
  std::vector<double> xc;
  elem_centroids_of(G, xc);
  const PartGeom geo{xc.data(), G->dim};
  bisect(ptr, adj, set, 0, nparts, p, weight, &geo);
  fh_copy_out(part, p);
  return 0;
  FH_GUARD_END("fh_mesh_partition_weighted")
}

extern "C" int fh_mesh_rank_elements(fh_mesh_t G, const int* part, int rank, int* n_owned, int* n_total, int* elems) {
  FH_GUARD_BEGIN
  FH_REQUIRE(G && part && n_owned && n_total, "fh_mesh_rank_elements: bad arguments");
  const int nl = G->nloc;
  std::vector<char> touched(G->nnode, 0);
  std::vector<int> own, ring;
  for (int e = 0; e < G->nel; e++)
    if (part[e] == rank) {
      own.push_back(e);
      for (int i = 0; i < nl; i++) touched[G->elem_dof[(size_t)e * nl + i]] = 1;
    }
  for (int e = 0; e < G->nel; e++) {
    if (part[e] == rank) continue;
    bool hit = false;
    for (int i = 0; i < nl && !hit; i++) hit = touched[G->elem_dof[(size_t)e * nl + i]];
    if (hit) ring.push_back(e);
  }
  *n_owned = (int)own.size();
  *n_total = (int)(own.size() + ring.size());
  if (elems) {
    std::copy(own.begin(), own.end(), elems);
    std::copy(ring.begin(), ring.end(), elems + own.size());
  }
  return 0;
  FH_GUARD_END("fh_mesh_rank_elements")
}

extern "C" int fh_mesh_submesh(fh_mesh_t G, int nsel, const int* sel, fh_mesh_t* out, int* node_gid) {
  FH_GUARD_BEGIN
  FH_REQUIRE(G && sel && out && nsel >= 1, "fh_mesh_submesh: bad arguments");
  const int nl = G->nloc, nf = nfaces_of(G->geom);
  std::unique_ptr<fh_mesh_s> m(new fh_mesh_s());
  m->geom = G->geom; m->dim = G->dim; m->nloc = nl; m->nel = nsel; m->level = G->level; m->amr_mode = G->amr_mode;
  std::vector<int> loc(G->nnode, -1), glob;
  m->elem_dof.resize((size_t)nsel * nl);
  m->face_flag.resize((size_t)nsel * nf);
  for (int k = 0; k < nsel; k++) {
    const int e = sel[k];
    FH_REQUIRE(e >= 0 && e < G->nel, "fh_mesh_submesh: element %d out of range", e);
    for (int i = 0; i < nl; i++) {
      const int g = G->elem_dof[(size_t)e * nl + i];
      if (loc[g] < 0) {
        loc[g] = (int)glob.size();
        glob.push_back(g);
      }
      m->elem_dof[(size_t)k * nl + i] = loc[g];
    }
    for (int f = 0; f < nf; f++) m->face_flag[(size_t)k * nf + f] = G->face_flag[(size_t)e * nf + f];     // cut faces are interior faces of G: never boundary
  }
  const int nn = (int)glob.size();
  m->coords.resize((size_t)nn * G->dim);
  for (int i = 0; i < nn; i++)
    for (int d = 0; d < G->dim; d++) m->coords[(size_t)i * G->dim + d] = G->coords[(size_t)glob[i] * G->dim + d];
  m->elem_level.resize(nsel);
  for (int k = 0; k < nsel; k++) m->elem_level[k] = G->elem_level.empty() ? G->level : G->elem_level[sel[k]];
  m->homogeneous = G->homogeneous;
  // first-touch renumbering, and the same permutation on the node map
  std::vector<int> before = m->elem_dof;
  first_touch_renumber(*m, nn);
  if (node_gid) {
    for (size_t q = 0; q < before.size(); q++) node_gid[m->elem_dof[q]] = glob[before[q]];
  }
  *out = m.release();
  return 0;
  FH_GUARD_END("fh_mesh_submesh")
}

extern "C" int fh_dd_topo_node_keys(fh_mesh_t G, const int* part, int nlevels, const fh_mesh_t* levels, const int* elem_gid0, int level, int64_t* gid,
                                    int* owner) {
  FH_GUARD_BEGIN
  FH_REQUIRE(G && part && levels && elem_gid0 && gid && owner && level >= 0 && level < nlevels && level <= 11, "fh_dd_topo_node_keys: bad arguments");
  const int geom = G->geom, dim = G->dim, nl = G->nloc, nch = nvert_of(geom);
  FH_REQUIRE(G->nnode < (1 << 24), "fh_dd_topo_node_keys: more than 2^24 coarse nodes");
  // lowest rank among the elements around every coarse node = owner of everything inside the entity that node stands for
  std::vector<int> minpart(G->nnode, INT32_MAX);
  for (int e = 0; e < G->nel; e++)
    for (int i = 0; i < nl; i++) {
      int& mp = minpart[G->elem_dof[(size_t)e * nl + i]];
      mp = std::min(mp, part[e]);
    }
  int node_at[3][3][3];                                  // local node with reference coordinates (sx-1, sy-1, sz-1)
  for (int i = 0; i < nl; i++) node_at[xc(geom, i, 0) + 1][xc(geom, i, 1) + 1][dim == 3 ? xc(geom, i, 2) + 1 : 1] = i;
  // every element of the level: its coarse element and the low corner of its box, in units of 2^-(level+1) of the coarse element
  const int N = 1 << (level + 1);
  struct Box { int ge, o[3], span; };
  std::vector<Box> box(levels[0]->nel);
  for (int e = 0; e < levels[0]->nel; e++) box[e] = Box{elem_gid0[e], {0, 0, 0}, N};
  for (int l = 0; l < level; l++) {
    const fh_mesh_s* mc = levels[l];
    FH_REQUIRE(!mc->child.empty() && (int)box.size() == mc->nel, "fh_dd_topo_node_keys: level %d has not been refined into level %d", l, l + 1);
    std::vector<Box> nb(levels[l + 1]->nel);
    for (int e = 0; e < mc->nel; e++) {
      if (!mc->refined[e]) {
        nb[mc->child[(size_t)e * nch]] = box[e];       // copied element of an adaptive level
        continue;
      }
      for (int j = 0; j < nch; j++) {                   // child j sits at the parent's vertex j
        Box b = box[e];
        b.span = box[e].span / 2;
        for (int d = 0; d < dim; d++) b.o[d] += (xc(geom, j, d) > 0) ? b.span : 0;
        nb[mc->child[(size_t)e * nch + j]] = b;
      }
    }
    box.swap(nb);
  }
  const fh_mesh_s* mf = levels[level];
  std::vector<char> done(mf->nnode, 0);
  for (int e = 0; e < mf->nel; e++) {
    const Box& b = box[e];
    const int* gn = &G->elem_dof[(size_t)b.ge * nl];
    for (int i = 0; i < nl; i++) {
      const int nd = mf->elem_dof[(size_t)e * nl + i];
      if (done[nd]) continue;
      done[nd] = 1;
      int p[3] = {0, 0, 0}, s[3] = {1, 1, 1};
      for (int d = 0; d < dim; d++) {
        p[d] = b.o[d] + (xc(geom, i, d) + 1) * b.span / 2;
        s[d] = p[d] == 0 ? 0 : p[d] == N ? 2 : 1;
      }
      const int ent = gn[node_at[s[0]][s[1]][s[2]]];
      int free_d[3], nfree = 0;
      for (int d = 0; d < dim; d++)
        if (s[d] == 1) free_d[nfree++] = d;
      int64_t t[3] = {0, 0, 0};
      auto corner = [&](int d1, int v1, int d2, int v2) {     // coarse node at the end(s) v of the free direction(s), the rest as s says
        int q[3] = {s[0], s[1], s[2]};
        q[d1] = v1;
        if (d2 >= 0) q[d2] = v2;
        return gn[node_at[q[0]][q[1]][q[2]]];
      };
      if (nfree == 1) {
        const int d = free_d[0];
        t[0] = corner(d, 0, -1, 0) < corner(d, 2, -1, 0) ? p[d] : N - p[d];
      } else if (nfree == 2) {
        const int d1 = free_d[0], d2 = free_d[1];
        int m1 = 0, m2 = 0, best = INT32_MAX;
        for (int a = 0; a < 2; a++)
          for (int c = 0; c < 2; c++) {
            const int g = corner(d1, 2 * a, d2, 2 * c);
            if (g < best) { best = g; m1 = a; m2 = c; }
          }
        const int64_t u = m1 == 0 ? p[d1] : N - p[d1], v = m2 == 0 ? p[d2] : N - p[d2];
        const bool first = corner(d1, 2 * (1 - m1), d2, 2 * m2) < corner(d1, 2 * m1, d2, 2 * (1 - m2));
        t[0] = first ? u : v;
        t[1] = first ? v : u;
      } else if (nfree == 3) {
        t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
      }
      gid[nd] = ((int64_t)ent << 39) | (t[0] << 26) | (t[1] << 13) | t[2];
      owner[nd] = minpart[ent];
    }
  }
  return 0;
  FH_GUARD_END("fh_dd_topo_node_keys")
}
