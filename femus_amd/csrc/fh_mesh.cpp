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
#include <unordered_map>

struct fh_mesh_s {
  int geom = 0, dim = 3, nloc = 27, nel = 0, nnode = 0, level = 0;
  int own[3] = {0, 0, 0};
  std::vector<int> elem_dof;      // [nel*nloc]
  std::vector<double> coords;     // [nnode*dim]
  std::vector<int> face_flag;     // [nel*nfaces]
  std::vector<int> child;         // [nel*nchild] (set by refine on the coarse mesh)
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
  FH_REQUIRE(nx > 0 && ny > 0 && nz >= 0 && out, "fh_mesh_box: bad arguments");
  fh_mesh_s* m = new fh_mesh_s();
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
  first_touch_renumber(*m, nnode);
  *out = m;
  return 0;
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

extern "C" int fh_mesh_refine(fh_mesh_t mc, fh_mesh_t* out) {
  FH_REQUIRE(mc && out, "fh_mesh_refine: null argument");
  const int geom = mc->geom, dim = mc->dim, nc = mc->nloc;
  const int nv = nvert_of(geom), ne = nedge_end_of(geom), nch = nv, nf = nfaces_of(geom);
  fh_mesh_s* m = new fh_mesh_s();
  m->geom = geom;
  m->dim = dim;
  m->nloc = nc;
  m->level = mc->level + 1;
  m->nel = mc->nel * nch;
  m->elem_dof.assign((size_t)m->nel * nc, -1);
  m->face_flag.assign((size_t)m->nel * nf, -1);
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
  // children: vertices and boundary flags (MeshRefinement.cpp:240-278)
  mc->child.resize((size_t)mc->nel * nch);
  for (int iel = 0; iel < mc->nel; iel++)
    for (int j = 0; j < nch; j++) {
      const int jel = iel * nch + j;
      mc->child[(size_t)iel * nch + j] = jel;
      for (int v = 0; v < nv; v++) m->elem_dof[(size_t)jel * nc + v] = mc->elem_dof[(size_t)iel * nc + f2c[j][v]];
      for (int f = 0; f < nf; f++) {
        int value = mc->face_flag[(size_t)iel * nf + f];
        if (value < -1 && child_on_face[f][j]) m->face_flag[(size_t)jel * nf + f] = value;
      }
    }
  int nnodes = mc->nnode;
  // edge mid-points (:356-417): first visit in (element, local edge) order creates the node
  {
    std::unordered_map<uint64_t, int> emap;
    emap.reserve((size_t)m->nel * 4);
    for (int iel = 0; iel < m->nel; iel++)
      for (int e = nv; e < ne; e++) {
        int a = m->elem_dof[(size_t)iel * nc + edge_v[e - nv][0]], b = m->elem_dof[(size_t)iel * nc + edge_v[e - nv][1]];
        if (a > b) std::swap(a, b);
        uint64_t key = ((uint64_t)a << 32) | (uint32_t)b;
        auto it = emap.find(key);
        if (it == emap.end()) it = emap.emplace(key, nnodes++).first;
        m->elem_dof[(size_t)iel * nc + e] = it->second;
      }
  }
  // quad-face centres of hexahedra (:526-561): (element, face 0..5) order
  if (geom == GEOM_HEX) {
    std::unordered_map<Key3, int, Key3Hash> fmap;
    fmap.reserve((size_t)m->nel * 4);
    for (int iel = 0; iel < m->nel; iel++)
      for (int f = 0; f < 6; f++) {
        int v[4];
        for (int k = 0; k < 4; k++) v[k] = m->elem_dof[(size_t)iel * nc + face_v[f][k]];
        std::sort(v, v + 4);
        Key3 key{v[0], v[1], v[2]};
        auto it = fmap.find(key);
        if (it == fmap.end()) it = fmap.emplace(key, nnodes++).first;
        m->elem_dof[(size_t)iel * nc + 20 + f] = it->second;
      }
  }
  // element centres (:598-616)
  for (int iel = 0; iel < m->nel; iel++) m->elem_dof[(size_t)iel * nc + nc - 1] = nnodes++;
  first_touch_renumber(*m, nnodes);
  // fine coordinates = biquadratic mesh prolongator x coarse coordinates (MeshRefinement.cpp:468-475);
  // rows are sums over the coarse element nodes in increasing global column order (CSR order)
  std::vector<double> EP;
  elem_prolongator(geom, FE_BIQUADRATIC, EP);
  m->coords.assign((size_t)m->nnode * dim, 0.0);
  std::vector<char> done(m->nnode, 0);
  std::vector<int> order(nc);
  for (int iel = 0; iel < mc->nel; iel++) {
    const int* cd = &mc->elem_dof[(size_t)iel * nc];
    for (int k = 0; k < nc; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cd[a] < cd[b]; });
    for (int j = 0; j < nch; j++) {
      const int jel = iel * nch + j;
      for (int i = 0; i < nc; i++) {
        const int row = m->elem_dof[(size_t)jel * nc + i];
        if (done[row]) continue;
        done[row] = 1;
        const double* pr = &EP[((size_t)j * nc + i) * nc];
        for (int d = 0; d < dim; d++) {
          double s = 0.0;
          for (int kk = 0; kk < nc; kk++) {
            const int k = order[kk];
            if (pr[k] != 0.0) s += pr[k] * mc->coords[(size_t)cd[k] * dim + d];
          }
          m->coords[(size_t)row * dim + d] = s;
        }
      }
    }
  }
  *out = m;
  return 0;
}

extern "C" int fh_mesh_clear_boundary_faces(fh_mesh_t m, unsigned face_mask) {
  const int nf = nfaces_of(m->geom);
  for (int iel = 0; iel < m->nel; iel++)
    for (int f = 0; f < nf; f++)
      if ((face_mask >> f) & 1u) m->face_flag[(size_t)iel * nf + f] = -1;
  return 0;
}

extern "C" int fh_mesh_set_coords(fh_mesh_t m, const double* coords) {
  FH_REQUIRE(m && coords, "fh_mesh_set_coords: null argument");
  memcpy(m->coords.data(), coords, m->coords.size() * sizeof(double));
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

extern "C" int fh_mesh_get(fh_mesh_t m, int* elem_dof, double* coords, int* face_flag) {
  if (elem_dof) memcpy(elem_dof, m->elem_dof.data(), m->elem_dof.size() * sizeof(int));
  if (coords) memcpy(coords, m->coords.data(), m->coords.size() * sizeof(double));
  if (face_flag) memcpy(face_flag, m->face_flag.data(), m->face_flag.size() * sizeof(int));
  return 0;
}

extern "C" int fh_mesh_child_elems(fh_mesh_t m, int* child) {
  FH_REQUIRE(!m->child.empty(), "fh_mesh_child_elems: mesh has not been refined");
  memcpy(child, m->child.data(), m->child.size() * sizeof(int));
  return 0;
}

static int mesh_ndofs(const fh_mesh_s* m, int fe) { return fe == FE_LINEAR ? m->own[0] : m->nnode; }

static void dirichlet_list(const fh_mesh_s* m, int fe, std::vector<int>& out) {
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
}

extern "C" int fh_mesh_dirichlet_dofs(fh_mesh_t m, int fe, int* n, int* dofs) {
  FH_REQUIRE(fe == 0 || fe == 2, "fh_mesh_dirichlet_dofs: fe must be 0 or 2");
  std::vector<int> list;
  dirichlet_list(m, fe, list);
  FH_REQUIRE(*n >= (int)list.size(), "fh_mesh_dirichlet_dofs: capacity %d < %d", *n, (int)list.size());
  *n = (int)list.size();
  memcpy(dofs, list.data(), list.size() * sizeof(int));
  return 0;
}

// a11: union of element couplings per row, sorted
extern "C" int fh_pattern_from_elements(int nel, int nloc, const int* elem_dof, int ndof, int* rowptr, int* col) {
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
  std::vector<int> buf;
  int64_t total = 0;
  if (!col) rowptr[0] = 0;
  for (int r = 0; r < ndof; r++) {
    buf.clear();
    for (int k = cnt[r]; k < cnt[r + 1]; k++) {
      const int* ed = elem_dof + (size_t)adj[k] * nloc;
      buf.insert(buf.end(), ed, ed + nloc);
    }
    std::sort(buf.begin(), buf.end());
    buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
    if (col) {
      FH_REQUIRE(rowptr[r] == total, "fh_pattern_from_elements: rowptr does not match (call with col=NULL first)");
      memcpy(col + total, buf.data(), buf.size() * sizeof(int));
    }
    total += (int64_t)buf.size();
    FH_REQUIRE(total < 2147483647ll, "fh_pattern_from_elements: nnz overflows int32");
    if (!col) rowptr[r + 1] = (int)total;
  }
  return 0;
}

// a14: P (fine x coarse), INSERT semantics (first insert wins; duplicates are identical rows)
extern "C" int fh_build_prolongator(fh_ctx_t ctx, fh_mesh_t mc, fh_mesh_t mf, int fe, int zero_bdc, fh_mat_t* out) {
  FH_REQUIRE(ctx && mc && mf && out, "fh_build_prolongator: null argument");
  FH_REQUIRE(fe == 0 || fe == 2, "fh_build_prolongator: fe must be 0 or 2");
  FH_REQUIRE(!mc->child.empty() && mf->nel == mc->nel * nvert_of(mc->geom), "fh_build_prolongator: fine is not the refinement of coarse");
  const int geom = mc->geom, nl = mc->nloc, nc = ndofs_of(geom, fe), nch = nvert_of(geom);
  const int nf = mesh_ndofs(mf, fe), ncc = mesh_ndofs(mc, fe);
  std::vector<double> EP;
  elem_prolongator(geom, fe, EP);
  std::vector<int> rowptr(nf + 1, 0);
  std::vector<char> done(nf, 0);
  // pass 1: row lengths
  for (int iel = 0; iel < mc->nel; iel++)
    for (int j = 0; j < nch; j++) {
      const int jel = mc->child[(size_t)iel * nch + j];
      for (int i = 0; i < nc; i++) {
        const int row = mf->elem_dof[(size_t)jel * nl + i];
        if (done[row]) continue;
        done[row] = 1;
        int cntr = 0;
        for (int k = 0; k < nc; k++) cntr += (EP[((size_t)j * nc + i) * nc + k] != 0.0);
        rowptr[row + 1] = cntr;
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
  for (int iel = 0; iel < mc->nel; iel++)
    for (int j = 0; j < nch; j++) {
      const int jel = mc->child[(size_t)iel * nch + j];
      for (int i = 0; i < nc; i++) {
        const int row = mf->elem_dof[(size_t)jel * nl + i];
        if (done[row]) continue;
        done[row] = 1;
        rowbuf.clear();
        for (int k = 0; k < nc; k++) {
          double v = EP[((size_t)j * nc + i) * nc + k];
          if (v == 0.0) continue;
          const int c = mc->elem_dof[(size_t)iel * nl + k];
          if (zero_bdc && (bf[row] || bc[c])) v = 0.0;   // pattern kept, value zeroed (mat_zero_rows keeps the pattern)
          rowbuf.emplace_back(c, v);
        }
        std::sort(rowbuf.begin(), rowbuf.end());
        int p = rowptr[row];
        for (auto& cv : rowbuf) {
          col[p] = cv.first;
          val[p++] = cv.second;
        }
      }
    }
  return fh_mat_create_csr(ctx, nf, ncc, rowptr.data(), col.data(), val.data(), out);
}
