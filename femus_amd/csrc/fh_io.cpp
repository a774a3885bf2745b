// Output and restart files behind the C-ABI -- host-side, what the applications call unconditionally after the
// solve (applications/001_Poisson/main.cpp:264-270):
//   fh_write_vtu        VTKWriter::Write(output_path, "biquadratic", vars) (src/07_mesh_or_solution/01_multiple_levels/01_output/
//                       VTKWriter.cpp:36-120, 460-770): one UnstructuredGrid piece, biquadratic cells (VTK types 28 / 29), Float32 points
//                       and point data, Int32 connectivity / offsets, UInt16 types, every DataArray "binary": base64(uint32 byte count)
//                       followed by base64(data), as print_data_array emits them; linear variables are carried to the biquadratic nodes
//                       (mean of the vertices a node sits between); VTK's 27-node hexahedron lists its face centres x-, x+, y-, y+, z-, z+
//                       (FEMuS: y-, x+, y+, x-, z-, z+), everything else coincides (Writer_one_level::FemusToVTKorToXDMFConn)
//   fh_vec_binary_print / fh_vec_binary_load   NumericVector::BinaryPrint / BinaryLoad (NumericVector.hpp:345-353; PetscVector: VecView /
//                       VecLoad on a binary viewer), the files MultiLevelSolution::SaveSolution / LoadSolution (MultiLevelSolution.cpp:
//                       1070-1126) write per variable: big-endian int32 class id 1211214, int32 length, float64 values (PETSc's
//                       documented binary Vec layout; PETSc itself is not under /root/reference)
#include "fh_internal.h"
#include "fh_fe.h"
#include <cstdio>
#include <cstring>
#include <memory>

using namespace fhfe;

int fh_mesh_host_arrays(fh_mesh_t m, int* dim, int* geom, int* nel, int* nnode, int* nloc, int* n_linear, const int** elem_dof, const double** coords);

static void b64_append(std::string& out, const unsigned char* p, size_t n) {
  static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  for (size_t i = 0; i < n; i += 3) {
    const unsigned b0 = p[i], b1 = i + 1 < n ? p[i + 1] : 0, b2 = i + 2 < n ? p[i + 2] : 0;
    out.push_back(T[b0 >> 2]);
    out.push_back(T[((b0 & 3) << 4) | (b1 >> 4)]);
    out.push_back(i + 1 < n ? T[((b1 & 15) << 2) | (b2 >> 6)] : '=');
    out.push_back(i + 2 < n ? T[b2 & 63] : '=');
  }
}

static std::string b64_array(const void* data, size_t bytes) {
  std::string out;
  const unsigned cnt = (unsigned)bytes;                       // the reference encodes the byte count on its own, then the data
  b64_append(out, reinterpret_cast<const unsigned char*>(&cnt), 4);
  b64_append(out, reinterpret_cast<const unsigned char*>(data), bytes);
  return out;
}

extern "C" int fh_write_vtu(fh_mesh_t mesh, const char* path, int nfields, const char* const* names, const int* fe, const double* const* values) {
  FH_REQUIRE(mesh && path && nfields >= 0 && (nfields == 0 || (names && fe && values)), "fh_write_vtu: bad arguments");
  int dim, geom, nel, nnode, nl, nlin;
  const int* ed;
  const double* xy;
  FH_TRY(fh_mesh_host_arrays(mesh, &dim, &geom, &nel, &nnode, &nl, &nlin, &ed, &xy));
  const int nv = nvert_of(geom);
  // VTK position -> FEMuS local node
  std::vector<int> order(nl);
  for (int i = 0; i < nl; i++) order[i] = i;
  if (geom == GEOM_HEX) {
    const int fx[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int f = 0; f < 6; f++)
      for (int n = 20; n < 26; n++)
        if (xc(geom, n, 0) == fx[f][0] && xc(geom, n, 1) == fx[f][1] && xc(geom, n, 2) == fx[f][2]) order[20 + f] = n;
  }
  std::vector<float> pts((size_t)nnode * 3, 0.f);
  for (int i = 0; i < nnode; i++)
    for (int d = 0; d < dim; d++) pts[(size_t)i * 3 + d] = (float)xy[(size_t)i * dim + d];
  std::vector<int> conn((size_t)nel * nl), offs(nel);
  std::vector<unsigned short> types(nel, (unsigned short)(geom == GEOM_HEX ? 29 : 28));
  for (int e = 0; e < nel; e++) {
    for (int k = 0; k < nl; k++) conn[(size_t)e * nl + k] = ed[(size_t)e * nl + order[k]];
    offs[e] = (e + 1) * nl;
  }
  for (int k = 0; k < nfields; k++)                          // names go into XML attributes as they are
    FH_REQUIRE(names[k] && !strpbrk(names[k], "\"&<>"), "fh_write_vtu: field %d: name with a character that an XML attribute cannot hold", k);
  FILE* f = fopen(path, "w");
  FH_REQUIRE(f != nullptr, "fh_write_vtu: cannot open %s", path);
  std::unique_ptr<FILE, int (*)(FILE*)> guard(f, fclose);
  fprintf(f, "<?xml version=\"1.0\"?>\n<VTKFile type = \"UnstructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n  <UnstructuredGrid>\n");
  fprintf(f, "    <Piece NumberOfPoints= \"%d\" NumberOfCells= \"%d\" >\n", nnode, nel);
  fprintf(f, "      <Points>\n        <DataArray type=\"Float32\" NumberOfComponents=\"3\" format=\"binary\">\n%s\n        </DataArray>\n      </Points>\n",
          b64_array(pts.data(), pts.size() * sizeof(float)).c_str());
  fprintf(f, "      <Cells>\n        <DataArray type=\"Int32\" Name=\"connectivity\" format=\"binary\">\n%s\n        </DataArray>\n",
          b64_array(conn.data(), conn.size() * sizeof(int)).c_str());
  fprintf(f, "        <DataArray type=\"Int32\" Name=\"offsets\" format=\"binary\">\n%s\n        </DataArray>\n", b64_array(offs.data(), offs.size() * sizeof(int)).c_str());
  fprintf(f, "        <DataArray type=\"UInt16\" Name=\"types\" format=\"binary\">\n%s\n        </DataArray>\n      </Cells>\n",
          b64_array(types.data(), types.size() * sizeof(unsigned short)).c_str());
  fprintf(f, "      <PointData Scalars=\"scalars\">\n");
  std::vector<float> fv(nnode);
  for (int k = 0; k < nfields; k++) {
    FH_REQUIRE(fe[k] == 0 || fe[k] == 2, "fh_write_vtu: field %d: fe must be 0 (linear) or 2 (biquadratic)", k);
    if (fe[k] == 2) {
      for (int i = 0; i < nnode; i++) fv[i] = (float)values[k][i];
    } else {
      // a Q1 field at every biquadratic node: mean of the vertices the node sits between (the element interpolation of the reference)
      std::vector<double> full(nnode, 0.0);
      for (int e = 0; e < nel; e++)
        for (int i = 0; i < nl; i++) {
          double s = 0.0;
          int cnt = 0;
          for (int v = 0; v < nv; v++) {
            bool on = true;
            for (int d = 0; d < dim; d++) on = on && (xc(geom, i, d) == 0 || xc(geom, i, d) == xc(geom, v, d));
            if (on) {
              s += values[k][ed[(size_t)e * nl + v]];
              cnt++;
            }
          }
          full[ed[(size_t)e * nl + i]] = s / cnt;
        }
      for (int i = 0; i < nnode; i++) fv[i] = (float)full[i];
    }
    fprintf(f, "        <DataArray type=\"Float32\" Name=\"%s\" format=\"binary\">\n%s\n        </DataArray>\n", names[k],
            b64_array(fv.data(), fv.size() * sizeof(float)).c_str());
  }
  fprintf(f, "      </PointData>\n    </Piece>\n  </UnstructuredGrid>\n</VTKFile>\n");
  const bool wrote = ferror(f) == 0 && fflush(f) == 0;       // a full disk shows up here or at the close
  const bool closed = fclose(guard.release()) == 0;
  FH_REQUIRE(wrote && closed, "fh_write_vtu: writing %s failed", path);
  return 0;
}

// GMVWriter::Write (src/07_mesh_or_solution/01_multiple_levels/01_output/GMVWriter.cpp:72-341), the second file 001_Poisson writes after
// the solve (main.cpp:267-269): binary GMV, "gmvinput" "ieeei4r8", 8-byte keywords, 32-bit counts, 64-bit values.  order 0 = "linear"
// (vertex nodes: phex8 / quad), anything else -- also "biquadratic" -- is the reference's QUADRATIC family (GMVWriter.cpp:102: vertex +
// edge nodes, phex20 / 8quad).  Nodes of that family are the first nvt mesh nodes (FEMuS numbers vertices, then edges, then faces /
// centres), coordinates and Lagrange variables are their nodal values (the Q_i -> Q_j projection at a node of both families is the value
// there; a linear variable at an edge node is the mean of the edge's vertices).  One rank: the METIS_DD cell variable is 0 everywhere.
// Keywords are written the way the reference does -- sprintf into one 10-byte buffer, 8 bytes out -- so that the bytes after a short
// keyword are what the previous keyword left there; variable names shorter than 8 characters are padded with zeros.
extern "C" int fh_write_gmv(fh_mesh_t mesh, const char* path, int order, int nfields, const char* const* names, const int* fe, const double* const* values) {
  FH_REQUIRE(mesh && path && nfields >= 0 && (nfields == 0 || (names && fe && values)), "fh_write_gmv: bad arguments");
  int dim, geom, nel, nnode, nl, nlin;
  const int* ed;
  const double* xy;
  FH_TRY(fh_mesh_host_arrays(mesh, &dim, &geom, &nel, &nnode, &nl, &nlin, &ed, &xy));
  const int index = order == 0 ? 0 : 1;
  const int nv = nvert_of(geom), nvq = nedge_end_of(geom);           // local nodes of the linear / quadratic family
  const int nloc_fam = index == 0 ? nv : nvq;
  // global count of the family: the largest node id an element lists among its first nloc_fam nodes, plus one
  int nvt_i = 0;
  for (int e = 0; e < nel; e++)
    for (int j = 0; j < nloc_fam; j++) nvt_i = std::max(nvt_i, ed[(size_t)e * nl + j] + 1);
  const unsigned nvt = (unsigned)nvt_i, nelu = (unsigned)nel;
  FILE* f = fopen(path, "wb");
  FH_REQUIRE(f != nullptr, "fh_write_gmv: cannot open %s", path);
  std::unique_ptr<FILE, int (*)(FILE*)> guard(f, fclose);
  char buffer[10] = {0};
  bool ok = true;
  auto key = [&](const char* w) {
    snprintf(buffer, sizeof(buffer), "%s", w);
    ok = ok && fwrite(buffer, 1, 8, f) == 8;
  };
  auto put = [&](const void* p, size_t bytes) { ok = ok && (bytes == 0 || fwrite(p, 1, bytes, f) == bytes); };
  key("gmvinput");
  key("ieeei4r8");
  key("nodes");
  put(&nvt, sizeof(unsigned));
  std::vector<double> v1(std::max<size_t>(nvt, (size_t)nel));
  for (int d = 0; d < 3; d++) {
    for (unsigned i = 0; i < nvt; i++) v1[i] = d < dim ? xy[(size_t)i * dim + d] : 0.0;
    put(v1.data(), nvt * sizeof(double));
  }
  key("cells");
  put(&nelu, sizeof(unsigned));
  const unsigned nvertices = (unsigned)nloc_fam;
  std::vector<unsigned> topo(nloc_fam);
  for (int e = 0; e < nel; e++) {
    if (geom == GEOM_HEX) key(index == 0 ? "phex8" : "phex20");
    else key(index == 0 ? "quad" : "8quad");
    put(&nvertices, sizeof(unsigned));
    for (int j = 0; j < nloc_fam; j++) topo[j] = (unsigned)ed[(size_t)e * nl + j] + 1u;
    put(topo.data(), topo.size() * sizeof(unsigned));
  }
  const unsigned zero = 0u, one = 1u;
  key("variable");
  key("METIS_DD");
  put(&zero, sizeof(unsigned));
  std::fill(v1.begin(), v1.begin() + nel, 0.0);
  put(v1.data(), (size_t)nel * sizeof(double));
  for (int k = 0; k < nfields; k++) {
    FH_REQUIRE(fe[k] == 0 || fe[k] == 2, "fh_write_gmv: field %d: fe must be 0 (linear) or 2 (biquadratic)", k);
    char name8[8] = {0};
    strncpy(name8, names[k], 8);
    put(name8, 8);
    put(&one, sizeof(unsigned));
    if (fe[k] == 2 || index == 0) {
      put(values[k], nvt * sizeof(double));                            // nodal values at the first nvt nodes
    } else {
      // a linear variable at the quadratic family's nodes: vertices keep their value, an edge node takes the mean of its two vertices
      std::vector<double> q(nvt, 0.0);
      for (int e = 0; e < nel; e++)
        for (int i = 0; i < nloc_fam; i++) {
          double s = 0.0;
          int cnt = 0;
          for (int v = 0; v < nv; v++) {
            bool on = true;
            for (int d = 0; d < dim; d++) on = on && (xc(geom, i, d) == 0 || xc(geom, i, d) == xc(geom, v, d));
            if (on) {
              s += values[k][ed[(size_t)e * nl + v]];
              cnt++;
            }
          }
          q[ed[(size_t)e * nl + i]] = s / cnt;
        }
      put(q.data(), nvt * sizeof(double));
    }
  }
  key("endvars");
  key("endgmv");
  FH_REQUIRE(ok, "fh_write_gmv: short write to %s", path);
  return 0;
}

static void put_be32(unsigned char* p, int v) {
  p[0] = (unsigned char)((unsigned)v >> 24), p[1] = (unsigned char)((unsigned)v >> 16), p[2] = (unsigned char)((unsigned)v >> 8), p[3] = (unsigned char)v;
}
static const int VEC_FILE_CLASSID = 1211214;

extern "C" int fh_host_binary_print(const char* path, int n, const double* values) {
  FH_REQUIRE(path && n >= 0 && (n == 0 || values), "fh_host_binary_print: bad arguments");
  FILE* f = fopen(path, "wb");
  FH_REQUIRE(f != nullptr, "fh_host_binary_print: cannot open %s", path);
  std::unique_ptr<FILE, int (*)(FILE*)> guard(f, fclose);
  std::vector<unsigned char> buf(8 + (size_t)n * 8);
  put_be32(&buf[0], VEC_FILE_CLASSID);
  put_be32(&buf[4], n);
  for (int i = 0; i < n; i++) {
    unsigned long long u;
    memcpy(&u, &values[i], 8);
    for (int b = 0; b < 8; b++) buf[8 + (size_t)i * 8 + b] = (unsigned char)(u >> (56 - 8 * b));
  }
  FH_REQUIRE(fwrite(buf.data(), 1, buf.size(), f) == buf.size(), "fh_host_binary_print: short write to %s", path);
  return 0;
}

// *n in: capacity of values (0 with values == NULL: query), out: length stored in the file
extern "C" int fh_host_binary_load(const char* path, int* n, double* values) {
  FH_REQUIRE(path && n, "fh_host_binary_load: bad arguments");
  FILE* f = fopen(path, "rb");
  FH_REQUIRE(f != nullptr, "Error: cannot locate file %s", path);
  std::unique_ptr<FILE, int (*)(FILE*)> guard(f, fclose);
  unsigned char h[8];
  FH_REQUIRE(fread(h, 1, 8, f) == 8, "fh_host_binary_load: %s is too short", path);
  const int cid = (int)(((unsigned)h[0] << 24) | ((unsigned)h[1] << 16) | ((unsigned)h[2] << 8) | h[3]);
  const int len = (int)(((unsigned)h[4] << 24) | ((unsigned)h[5] << 16) | ((unsigned)h[6] << 8) | h[7]);
  FH_REQUIRE(cid == VEC_FILE_CLASSID && len >= 0, "%s is not a binary vector file", path);
  if (!values) {
    *n = len;
    return 0;
  }
  FH_REQUIRE(*n >= len, "fh_host_binary_load: %s holds %d values, capacity %d", path, len, *n);
  std::vector<unsigned char> buf((size_t)len * 8);
  FH_REQUIRE(fread(buf.data(), 1, buf.size(), f) == buf.size(), "fh_host_binary_load: %s is truncated", path);
  for (int i = 0; i < len; i++) {
    unsigned long long u = 0;
    for (int b = 0; b < 8; b++) u = (u << 8) | buf[(size_t)i * 8 + b];
    memcpy(&values[i], &u, 8);
  }
  *n = len;
  return 0;
}

extern "C" int fh_vec_binary_print(fh_vec_t v, const char* path) {
  FH_REQUIRE(v && path, "fh_vec_binary_print: null argument");
  std::vector<double> h(v->n_local);
  FH_TRY(fh_vec_download(v, h.data()));
  return fh_host_binary_print(path, v->n_local, h.data());
}

extern "C" int fh_vec_binary_load(fh_vec_t v, const char* path) {
  FH_REQUIRE(v && path, "fh_vec_binary_load: null argument");
  int n = 0;
  FH_TRY(fh_host_binary_load(path, &n, nullptr));
  FH_REQUIRE(n == v->n_local, "fh_vec_binary_load: %s holds %d values, the vector %d", path, n, v->n_local);
  std::vector<double> h(n);
  FH_TRY(fh_host_binary_load(path, &n, h.data()));
  return fh_vec_upload(v, h.data());
}
