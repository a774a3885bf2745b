// Mesh refinement ON THE DEVICE (round 5) and the device-resident copy of a mesh that the set-up calls after it read instead of uploading
// the element table again (fh_mat_create_from_mesh, fh_assembler_create_mesh, fh_build_prolongator).
//
// MeshRefinement::RefineMesh (MeshRefinement.cpp:197-493) + Mesh::... node renumbering (Mesh.cpp:517-559), nprocs = 1, as fh_mesh.cpp
// restates them on the host.  The host loop numbers new nodes in visiting order and then renumbers everything by FIRST TOUCH in three
// passes (vertices, edge mid-points, face centres + element centres; element by element, local node by local node).  The second numbering
// makes the first one irrelevant: the final id of a node is the rank, among all first touches, of its first touch in the order
// (class, element, local node).  That order is an integer `occ`, so:
//   1. every (fine element, local node) names the node it touches -- a coarse node id, an edge (two vertex ids), a quadrilateral face
//      (its smallest vertex id and the vertex diagonal to it), or the element itself for its centre; edges and faces get a slot of an
//      open-addressing table (atomicCAS on a 64-bit key) -- and lowers first[node] to its occ with atomicMin;
//   2. flag[occ] = (first[node] == occ); an exclusive scan of the flags is the new numbering;
//   3. every (element, local node) reads its id, and the first touch of a node also writes its coordinates: the row of the biquadratic
//      element prolongator times the coarse coordinates, products and sums rounded separately and added in increasing coarse node id
//      (MeshRefinement.cpp:468-475: a CSR row of the mesh prolongator times the coarse vector) -- the bits of the host path.
// Slot numbers depend on the race, node ids do not: the result is the host's, bit for bit (tests/test_gpu_mesh_device.py).
#include "fh_internal.h"
#include <algorithm>
#include <vector>

void fh_meshdev_free(fh_mesh_dev* d) {
  if (!d) return;
  for (void* q : {(void*)d->d_elem_dof, (void*)d->d_coords, (void*)d->d_face_flag, (void*)d->d_elem_level, (void*)d->d_child, (void*)d->d_refined})
    if (q) hipFree(q);
  delete d;
}

int fh_meshdev_upload(fh_ctx_t ctx, int nel, int nnode, int nloc, int dim, int nf, const int* elem_dof, const double* coords, const int* face_flag,
                      const int* elem_level, fh_mesh_dev** out) {
  fh_mesh_dev* d = new fh_mesh_dev();
  d->ctx = ctx;
  d->nel = nel; d->nnode = nnode; d->nloc = nloc; d->dim = dim; d->nf = nf;
  auto up = [&](void** p, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(p, std::max<size_t>(bytes, 8)));
    if (bytes && h) FH_CHECK_HIP(hipMemcpyAsync(*p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
  };
  int rc = up((void**)&d->d_elem_dof, elem_dof, (size_t)nel * nloc * sizeof(int)) || up((void**)&d->d_coords, coords, (size_t)nnode * dim * sizeof(double)) ||
           up((void**)&d->d_face_flag, face_flag, (size_t)nel * nf * sizeof(int)) || up((void**)&d->d_elem_level, elem_level, (size_t)nel * sizeof(int));
  if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = 2;
  if (rc) {
    fh_meshdev_free(d);
    return 2;
  }
  *out = d;
  return 0;
}

// ---- exclusive scan of n ints (out[n] = total); in and out may be the same array ---------------------------------------------------------
constexpr int SC_T = 256, SC_E = 8, SC_B = SC_T * SC_E;
__device__ __forceinline__ int sc_block_inclusive(int v, int* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = 1; off < SC_T; off <<= 1) {
    const int a = t >= off ? sh[t - off] : 0;
    __syncthreads();
    sh[t] += a;
    __syncthreads();
  }
  return sh[t];
}
__global__ __launch_bounds__(SC_T) void k_scan_local(const int* in, int* out, int n, int* __restrict__ bsum) {
  __shared__ int sh[SC_T];
  const size_t base = (size_t)blockIdx.x * SC_B + (size_t)threadIdx.x * SC_E;
  int v[SC_E], s = 0;
  for (int e = 0; e < SC_E; e++) {
    v[e] = (base + e < (size_t)n) ? in[base + e] : 0;
    s += v[e];
  }
  const int incl = sc_block_inclusive(s, sh);
  int run = incl - s;
  for (int e = 0; e < SC_E; e++) {
    if (base + e < (size_t)n) out[base + e] = run;
    run += v[e];
  }
  if (threadIdx.x == SC_T - 1) bsum[blockIdx.x] = incl;
}
__global__ __launch_bounds__(SC_T) void k_scan_bsums(int* __restrict__ bsum, int nb, int* __restrict__ total) {
  __shared__ int sh[SC_T];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += SC_T) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    const int incl = sc_block_inclusive(v, sh);
    if (i < nb) bsum[i] = incl - v + carry;
    __syncthreads();
    if (threadIdx.x == SC_T - 1) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(SC_T) void k_scan_add(int* __restrict__ out, int n, const int* __restrict__ bsum) {
  const int add = bsum[blockIdx.x];
  const size_t base = (size_t)blockIdx.x * SC_B + threadIdx.x;
  for (int e = 0; e < SC_E; e++) {
    const size_t i = base + (size_t)e * SC_T;
    if (i < (size_t)n) out[i] += add;
  }
}
// out must hold n + 1 ints; d_bsum at least n / SC_B + 1
static int device_exclusive_scan(hipStream_t st, const int* d_in, int* d_out, int n, int* d_bsum) {
  const int nb = std::max(1, (n + SC_B - 1) / SC_B);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SC_T), 0, st, d_in, d_out, n, d_bsum);
  hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SC_T), 0, st, d_bsum, nb, d_out + n);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SC_T), 0, st, d_out, n, d_bsum);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- refinement kernels ----------------------------------------------------------------------------------------------------------------------
struct RfTab {          // reference-element tables of fh_mesh.cpp (refine_tables)
  int nv, ne, nc, nch, nf, dim;
  int f2c[8][8], edge_v[12][2], face_v[6][4], face_diag[6][4];
  unsigned char cof[6][8];
};
constexpr unsigned long long RF_EMPTY = ~0ull;
constexpr int RF_NONE = 0x7f7f7f7f;

__global__ __launch_bounds__(256) void k_rf_mark(int nel, int level, const int* __restrict__ lvl, const unsigned char* __restrict__ flags, char* __restrict__ refined,
                                                 int* __restrict__ cnt, int nch) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nel) return;
  const bool r = lvl[e] == level && (!flags || flags[e]);
  refined[e] = r;
  cnt[e] = r ? nch : 1;
}

__global__ __launch_bounds__(256) void k_rf_children(RfTab T, int nel_c, const char* __restrict__ refined, const int* __restrict__ start, const int* __restrict__ ed_c,
                                                     const int* __restrict__ ff_c, const int* __restrict__ lvl_c, int* __restrict__ child, int* __restrict__ ed_f,
                                                     int* __restrict__ ff_f, int* __restrict__ lvl_f, int* __restrict__ parent) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)nel_c * T.nch) return;
  const int iel = (int)(t / T.nch), j = (int)(t % T.nch);
  const int* cd = ed_c + (size_t)iel * T.nc;
  const int* cf = ff_c + (size_t)iel * T.nf;
  if (refined[iel]) {
    const int jel = start[iel] + j;
    child[t] = jel;
    lvl_f[jel] = lvl_c[iel] + 1;
    parent[jel] = iel * T.nch + j;
    for (int v = 0; v < T.nv; v++) ed_f[(size_t)jel * T.nc + v] = cd[T.f2c[j][v]];
    for (int f = 0; f < T.nf; f++) {
      const int value = cf[f];
      ff_f[(size_t)jel * T.nf + f] = (value < -1 && T.cof[f][j]) ? value : -1;
    }
  } else if (j == 0) {
    const int jel = start[iel];
    child[t] = jel;
    lvl_f[jel] = lvl_c[iel];
    parent[jel] = -(iel + 1);
    for (int i = 0; i < T.nc; i++) ed_f[(size_t)jel * T.nc + i] = cd[i];
    for (int f = 0; f < T.nf; f++) ff_f[(size_t)jel * T.nf + f] = cf[f] < -1 ? cf[f] : -1;
  } else {
    child[t] = -1;
  }
}

__device__ __forceinline__ int rf_insert(unsigned long long* keys, unsigned mask, int shift, unsigned long long key) {
  unsigned s = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> shift) & mask;
  for (;;) {
    unsigned long long old = keys[s];
    if (old == key) return (int)s;
    if (old == RF_EMPTY) {
      old = atomicCAS(&keys[s], RF_EMPTY, key);
      if (old == RF_EMPTY || old == key) return (int)s;
    }
    s = (s + 1) & mask;
  }
}
__device__ __forceinline__ int rf_occ(const RfTab& T, int nel_f, int jel, int i) {
  if (i < T.nv) return jel * T.nv + i;
  if (i < T.ne) return nel_f * T.nv + jel * (T.ne - T.nv) + (i - T.nv);
  return nel_f * T.ne + jel * (T.nc - T.ne) + (i - T.ne);
}

__global__ __launch_bounds__(256) void k_rf_touch(RfTab T, int nel_f, const int* __restrict__ parent, const int* __restrict__ ed_f, unsigned long long* __restrict__ keys,
                                                  unsigned mask, int shift, int E0, int C0, int* __restrict__ first, int* __restrict__ ident) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)nel_f * T.nc) return;
  const int jel = (int)(t / T.nc), i = (int)(t % T.nc);
  const int* fd = ed_f + (size_t)jel * T.nc;
  int id;
  if (parent[jel] < 0 || i < T.nv) {
    id = fd[i];
  } else if (i < T.ne) {
    int a = fd[T.edge_v[i - T.nv][0]], b = fd[T.edge_v[i - T.nv][1]];
    if (a > b) { const int c = a; a = b; b = c; }
    id = E0 + rf_insert(keys, mask, shift, ((unsigned long long)(unsigned)a << 32) | (unsigned)b);
  } else if (i < T.nc - 1) {
    const int f = i - T.ne;
    int v[4], km = 0;
    for (int k = 0; k < 4; k++) v[k] = fd[T.face_v[f][k]];
    for (int k = 1; k < 4; k++)
      if (v[k] < v[km]) km = k;
    id = E0 + rf_insert(keys, mask, shift, (1ull << 63) | ((unsigned long long)(unsigned)v[km] << 32) | (unsigned)v[T.face_diag[f][km]]);
  } else {
    id = C0 + jel;
  }
  ident[t] = id;
  atomicMin(&first[id], rf_occ(T, nel_f, jel, i));
}

__global__ __launch_bounds__(256) void k_rf_flag(RfTab T, int nel_f, const int* __restrict__ first, const int* __restrict__ ident, int* __restrict__ flag) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)nel_f * T.nc) return;
  const int occ = rf_occ(T, nel_f, (int)(t / T.nc), (int)(t % T.nc));
  flag[occ] = first[ident[t]] == occ;
}

__global__ __launch_bounds__(256) void k_rf_number(RfTab T, int nel_f, const int* __restrict__ parent, const int* __restrict__ first, const int* __restrict__ ident,
                                                   const int* __restrict__ pos, const int* __restrict__ ed_c, const double* __restrict__ xc, const int* __restrict__ cnt,
                                                   const int* __restrict__ nzk, const double* __restrict__ EP, int* __restrict__ ed_f, double* __restrict__ xf) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)nel_f * T.nc) return;
  const int jel = (int)(t / T.nc), i = (int)(t % T.nc);
  const int fo = first[ident[t]];
  const int id = pos[fo];
  ed_f[t] = id;
  if (fo != rf_occ(T, nel_f, jel, i)) return;
  const int par = parent[jel];
  double s[3] = {0.0, 0.0, 0.0};
  if (par < 0) {
    const int c = ed_c[(size_t)(-par - 1) * T.nc + i];
    for (int d = 0; d < T.dim; d++) s[d] = xc[(size_t)c * T.dim + d];
  } else {
    const int iel = par / T.nch, row = (par % T.nch) * T.nc + i;
    const int* cd = ed_c + (size_t)iel * T.nc;
    const int n = cnt[row];
    const int* nz = nzk + (size_t)row * T.nc;
    const double* pr = EP + (size_t)row * T.nc;
    int last = -1;
    for (int a = 0; a < n; a++) {
      int best = 0x7fffffff, bk = 0;
      for (int b = 0; b < n; b++) {
        const int k = nz[b], c = cd[k];
        if (c > last && c < best) { best = c; bk = k; }
      }
      last = best;
      const double w = pr[bk];
      for (int d = 0; d < T.dim; d++) s[d] = __dadd_rn(s[d], __dmul_rn(w, xc[(size_t)best * T.dim + d]));
    }
  }
  for (int d = 0; d < T.dim; d++) xf[(size_t)id * T.dim + d] = s[d];
}

namespace {
struct Scratch {
  std::vector<void*> p;
  ~Scratch() {
    for (void* q : p)
      if (q) hipFree(q);
  }
  template <class T>
  int get(T** out, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 2) * sizeof(T)) != hipSuccess) {
      fh_set_error("fh_mesh_refine_device: out of device memory");
      return 2;
    }
    p.push_back(q);
    *out = (T*)q;
    return 0;
  }
};
}   // namespace

int fh_meshdev_refine(fh_ctx_t ctx, const fh_refine_tables& H, fh_mesh_dev* C, int level_c, const unsigned char* flags, fh_refine_result* R) {
  hipStream_t st = ctx->stream;
  RfTab T;
  memset(&T, 0, sizeof(T));
  T.nv = H.nv; T.ne = H.ne; T.nc = H.nc; T.nch = H.nch; T.nf = H.nf; T.dim = H.dim;
  memcpy(T.f2c, H.f2c, sizeof(T.f2c));
  memcpy(T.edge_v, H.edge_v, sizeof(T.edge_v));
  memcpy(T.face_v, H.face_v, sizeof(T.face_v));
  memcpy(T.face_diag, H.face_diag, sizeof(T.face_diag));
  memcpy(T.cof, H.cof, sizeof(T.cof));
  const int nel_c = C->nel, nc = T.nc, nch = T.nch;
  FH_REQUIRE(C->nloc == nc && C->dim == T.dim && C->nf == T.nf, "fh_mesh_refine_device: the device copy of the mesh does not match its geometry");
  Scratch B;
  int *d_cnt, *d_start, *d_bsum, *d_tab_cnt, *d_tab_nz;
  unsigned char* d_flags = nullptr;
  double* d_EP;
  const size_t nslot = (size_t)nel_c * nch;
  if (B.get(&d_cnt, (size_t)nel_c) || B.get(&d_start, (size_t)nel_c + 1) || B.get(&d_bsum, nslot * nc / SC_B + 2) || B.get(&d_tab_cnt, H.cnt.size()) ||
      B.get(&d_tab_nz, H.nzk.size()) || B.get(&d_EP, H.EP.size()))
    return 2;
  if (flags) {
    if (B.get(&d_flags, (size_t)nel_c)) return 2;
    FH_CHECK_HIP(hipMemcpyAsync(d_flags, flags, (size_t)nel_c, hipMemcpyHostToDevice, st));
  }
  FH_CHECK_HIP(hipMemcpyAsync(d_tab_cnt, H.cnt.data(), H.cnt.size() * sizeof(int), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemcpyAsync(d_tab_nz, H.nzk.data(), H.nzk.size() * sizeof(int), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemcpyAsync(d_EP, H.EP.data(), H.EP.size() * sizeof(double), hipMemcpyHostToDevice, st));
  // which elements are split and where their children are: handed to the coarse mesh's device copy when everything below has succeeded
  int* d_child = nullptr;
  char* d_refined = nullptr;
  struct ChildGuard {
    int*& c;
    char*& r;
    ~ChildGuard() {
      if (c) hipFree(c);
      if (r) hipFree(r);
    }
  } child_guard{d_child, d_refined};
  FH_CHECK_HIP(hipMalloc(&d_child, std::max<size_t>(nslot, 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_refined, std::max<size_t>(nel_c, 8)));
  if (nel_c) hipLaunchKernelGGL(k_rf_mark, dim3(fh_div_up(nel_c, 256)), dim3(256), 0, st, nel_c, level_c, C->d_elem_level, d_flags, d_refined, d_cnt, nch);
  FH_TRY(device_exclusive_scan(st, d_cnt, d_start, nel_c, d_bsum));
  int nel_f = 0;
  FH_CHECK_HIP(hipMemcpyAsync(&nel_f, d_start + nel_c, sizeof(int), hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipStreamSynchronize(st));
  FH_REQUIRE((int64_t)nel_f * nc < (int64_t)RF_NONE, "fh_mesh_refine_device: %d fine elements do not fit 32-bit ids", nel_f);
  const size_t nocc = (size_t)nel_f * nc;
  fh_mesh_dev* F = new fh_mesh_dev();
  F->ctx = ctx;
  F->nel = nel_f; F->nloc = nc; F->dim = T.dim; F->nf = T.nf;
  struct Guard {
    fh_mesh_dev* f;
    ~Guard() { fh_meshdev_free(f); }
  } guard{F};
  FH_CHECK_HIP(hipMalloc(&F->d_elem_dof, std::max<size_t>(nocc, 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&F->d_face_flag, std::max<size_t>((size_t)nel_f * T.nf, 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&F->d_elem_level, std::max<size_t>(nel_f, 2) * sizeof(int)));
  // edge / face table: every key fits with load <= 1/2 even if no edge or face were shared
  const size_t nkeys = (size_t)nel_f * (size_t)(nc - 1 - T.nv);
  int log2cap = 6;
  while (((size_t)1 << log2cap) < 2 * nkeys) log2cap++;
  const size_t cap = (size_t)1 << log2cap;
  const int E0 = C->nnode, C0 = E0 + (int)cap;
  FH_REQUIRE((int64_t)C->nnode + (int64_t)cap + nel_f < (int64_t)RF_NONE, "fh_mesh_refine_device: the node table does not fit 32-bit ids");
  const size_t nident = (size_t)C0 + nel_f;
  int *d_parent, *d_first, *d_ident, *d_flag;
  unsigned long long* d_keys;
  if (B.get(&d_parent, (size_t)nel_f) || B.get(&d_first, nident) || B.get(&d_ident, nocc) || B.get(&d_flag, nocc + 1) || B.get(&d_keys, cap)) return 2;
  FH_CHECK_HIP(hipMemsetAsync(d_keys, 0xFF, cap * sizeof(unsigned long long), st));
  FH_CHECK_HIP(hipMemsetAsync(d_first, 0x7f, nident * sizeof(int), st));
  if (nslot)
    hipLaunchKernelGGL(k_rf_children, dim3((unsigned)((nslot + 255) / 256)), dim3(256), 0, st, T, nel_c, d_refined, d_start, C->d_elem_dof, C->d_face_flag,
                       C->d_elem_level, d_child, F->d_elem_dof, F->d_face_flag, F->d_elem_level, d_parent);
  const unsigned gb = (unsigned)((nocc + 255) / 256);
  if (nocc) {
    hipLaunchKernelGGL(k_rf_touch, dim3(gb), dim3(256), 0, st, T, nel_f, d_parent, F->d_elem_dof, d_keys, (unsigned)(cap - 1), 64 - log2cap, E0, C0, d_first, d_ident);
    hipLaunchKernelGGL(k_rf_flag, dim3(gb), dim3(256), 0, st, T, nel_f, d_first, d_ident, d_flag);
  }
  FH_TRY(device_exclusive_scan(st, d_flag, d_flag, (int)nocc, d_bsum));
  int own[3] = {0, 0, 0};
  FH_CHECK_HIP(hipMemcpyAsync(&own[0], d_flag + (size_t)nel_f * T.nv, sizeof(int), hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipMemcpyAsync(&own[1], d_flag + (size_t)nel_f * T.ne, sizeof(int), hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipMemcpyAsync(&own[2], d_flag + nocc, sizeof(int), hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipStreamSynchronize(st));
  const int nnode_f = own[2];
  F->nnode = nnode_f;
  FH_CHECK_HIP(hipMalloc(&F->d_coords, std::max<size_t>((size_t)nnode_f * T.dim, 2) * sizeof(double)));
  if (nocc)
    hipLaunchKernelGGL(k_rf_number, dim3(gb), dim3(256), 0, st, T, nel_f, d_parent, d_first, d_ident, d_flag, C->d_elem_dof, C->d_coords, d_tab_cnt, d_tab_nz, d_EP,
                       F->d_elem_dof, F->d_coords);
  FH_CHECK_HIP(hipGetLastError());
  // the host copies every host-side consumer of the mesh reads (boundary lists, hanging-node constraints, partitioner, writers)
  R->nel = nel_f;
  R->nnode = nnode_f;
  for (int k = 0; k < 3; k++) R->own[k] = own[k];
  R->elem_dof.resize(nocc);
  R->coords.resize((size_t)nnode_f * T.dim);
  R->face_flag.resize((size_t)nel_f * T.nf);
  R->elem_level.resize(nel_f);
  R->child.resize(nslot);
  R->refined.resize(nel_c);
  if (nocc) FH_CHECK_HIP(hipMemcpyAsync(R->elem_dof.data(), F->d_elem_dof, nocc * sizeof(int), hipMemcpyDeviceToHost, st));
  if (nnode_f) FH_CHECK_HIP(hipMemcpyAsync(R->coords.data(), F->d_coords, R->coords.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  if (nel_f) FH_CHECK_HIP(hipMemcpyAsync(R->face_flag.data(), F->d_face_flag, R->face_flag.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  if (nel_f) FH_CHECK_HIP(hipMemcpyAsync(R->elem_level.data(), F->d_elem_level, (size_t)nel_f * sizeof(int), hipMemcpyDeviceToHost, st));
  if (nslot) FH_CHECK_HIP(hipMemcpyAsync(R->child.data(), d_child, nslot * sizeof(int), hipMemcpyDeviceToHost, st));
  if (nel_c) FH_CHECK_HIP(hipMemcpyAsync(R->refined.data(), d_refined, (size_t)nel_c, hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipStreamSynchronize(st));
  guard.f = nullptr;
  R->dev = F;
  if (C->d_child) hipFree(C->d_child);
  if (C->d_refined) hipFree(C->d_refined);
  C->d_child = d_child;
  C->d_refined = d_refined;
  d_child = nullptr;
  d_refined = nullptr;
  return 0;
}
