// XDMF + HDF5 output behind the C-ABI: XDMFWriter::Write (src/07_mesh_or_solution/01_multiple_levels/01_output/
// XDMFWriter.cpp:103-445), the biquadratic ("biquadratic" order) form the applications ask for:
//   <prefix>.level<L>.<step>.biquadratic.xmf   one Grid "Mesh": Topology Hexahedron_27 / Quadrilateral_9 -> :/CONNECTIVITY, Geometry X_Y_Z ->
//                                              :/NODES_X1..3, the cell attribute Domain_partitions, one node attribute per variable
//   <prefix>.level<L>.<step>.biquadratic.h5    /NODES_X1, /NODES_X2, /NODES_X3 (nvt x 1 doubles), /CONNECTIVITY (nel * ndofs x 1 ints, local nodes
//                                              in Writer_one_level::FemusToVTKorToXDMFConn order), /DOMAIN_PARTITIONS (nel x 1 doubles), /<name>
// HDF5 is not a link-time dependency of the library: libhdf5 is opened with dlopen at the first call (the image has 1.10.6 under /opt/conda/lib)
// and the dozen entry points used are declared here with the 1.10 ABI (hid_t = int64_t); without the library the call returns an error.
#include "fh_internal.h"
#include "fh_fe.h"
#include <cstdio>
#include <dlfcn.h>
#include <memory>
#include <string>

using namespace fhfe;

int fh_mesh_host_arrays(fh_mesh_t m, int* dim, int* geom, int* nel, int* nnode, int* nloc, int* n_linear, const int** elem_dof, const double** coords);

namespace {
typedef int64_t hid_t;
typedef unsigned long long hsize_t;
typedef int herr_t;
struct Hdf5 {
  void* lib = nullptr;
  herr_t (*H5open)() = nullptr;
  hid_t (*H5Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
  herr_t (*H5Fclose)(hid_t) = nullptr;
  hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
  herr_t (*H5Sclose)(hid_t) = nullptr;
  hid_t (*H5Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
  herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
  herr_t (*H5Dclose)(hid_t) = nullptr;
  hid_t* native_double = nullptr;
  hid_t* native_int = nullptr;
  bool ok = false;
};
Hdf5& hdf5() {
  static Hdf5 h;
  static bool tried = false;
  if (tried) return h;
  tried = true;
  const char* names[] = {getenv("FEMUS_HIP_HDF5"), "libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    h.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h.lib) break;
  }
  if (!h.lib) return h;
#define FH_SYM(name) *(void**)(&h.name) = dlsym(h.lib, #name)
  FH_SYM(H5open); FH_SYM(H5Fcreate); FH_SYM(H5Fclose); FH_SYM(H5Screate_simple); FH_SYM(H5Sclose); FH_SYM(H5Dcreate2); FH_SYM(H5Dwrite); FH_SYM(H5Dclose);
#undef FH_SYM
  h.native_double = (hid_t*)dlsym(h.lib, "H5T_NATIVE_DOUBLE_g");
  h.native_int = (hid_t*)dlsym(h.lib, "H5T_NATIVE_INT_g");
  h.ok = h.H5open && h.H5Fcreate && h.H5Fclose && h.H5Screate_simple && h.H5Sclose && h.H5Dcreate2 && h.H5Dwrite && h.H5Dclose && h.native_double && h.native_int;
  if (h.ok) h.ok = h.H5open() >= 0;
  return h;
}
int put(Hdf5& h, hid_t file, const char* name, hid_t type, hsize_t rows, const void* data) {
  const hsize_t dims[2] = {rows, 1};
  const hid_t sp = h.H5Screate_simple(2, dims, nullptr);
  if (sp < 0) return 1;
  const hid_t ds = h.H5Dcreate2(file, name, type, sp, 0, 0, 0);
  int bad = ds < 0;
  if (!bad) {
    bad |= h.H5Dwrite(ds, type, 0, 0, 0, data) < 0;
    bad |= h.H5Dclose(ds) < 0;
  }
  bad |= h.H5Sclose(sp) < 0;
  return bad;
}
}  // namespace

extern "C" int fh_xdmf_available() { return hdf5().ok ? 1 : 0; }

// output_path / prefix / level / time_step as XDMFWriter::Write takes them; values[k]: nodal array of the biquadratic (fe 2) or linear (fe 0)
// family (linear fields are carried to the biquadratic nodes like in fh_write_vtu)
extern "C" int fh_write_xdmf(fh_mesh_t mesh, const char* output_path, const char* prefix, int level, int time_step, int nfields, const char* const* names,
                             const int* fe, const double* const* values) {
  FH_GUARD_BEGIN
  FH_REQUIRE(mesh && output_path && prefix && nfields >= 0 && (nfields == 0 || (names && fe && values)), "fh_write_xdmf: bad arguments");
  Hdf5& h = hdf5();
  FH_REQUIRE(h.ok, "fh_write_xdmf: no usable HDF5 library (libhdf5.so, 1.10 ABI; FEMUS_HIP_HDF5 names another file)");
  int dim, geom, nel, nnode, nl, nlin;
  const int* ed;
  const double* xy;
  FH_TRY(fh_mesh_host_arrays(mesh, &dim, &geom, &nel, &nnode, &nl, &nlin, &ed, &xy));
  for (int k = 0; k < nfields; k++) {
    FH_REQUIRE(fe[k] == 0 || fe[k] == 2, "fh_write_xdmf: field %d: fe must be 0 (linear) or 2 (biquadratic)", k);
    for (const char* q = names[k]; *q; q++) FH_REQUIRE(*q != '"' && *q != '<' && *q != '&' && *q != '/', "fh_write_xdmf: field name %s cannot be a dataset / attribute name", names[k]);
  }
  const std::string stem = std::string(prefix) + ".level" + std::to_string(level) + "." + std::to_string(time_step) + ".biquadratic";
  const std::string xmf = std::string(output_path) + "/" + stem + ".xmf", h5name = stem + ".h5", h5path = std::string(output_path) + "/" + h5name;
  // ---- light data ----
  {
    FILE* f = fopen(xmf.c_str(), "w");
    FH_REQUIRE(f != nullptr, "fh_write_xdmf: cannot open %s", xmf.c_str());
    bool ok = true;
    auto P = [&](const std::string& s2) { ok = ok && fputs(s2.c_str(), f) >= 0; };
    const std::string nvt = std::to_string(nnode), ne = std::to_string(nel);
    P("<?xml version=\"1.0\" ?>\n<!DOCTYPE Xdmf SYSTEM \"Xdmf.dtd []\">\n<Xdmf>\n<Domain>\n<Grid Name=\"Mesh\">\n");
    P("<Time Value =\"" + std::to_string(time_step) + "\" />\n");
    P(std::string("<Topology Type=\"") + (geom == GEOM_HEX ? "Hexahedron_27" : "Quadrilateral_9") + "\" Dimensions=\"" + ne + "\">\n");
    P("<DataStructure DataType=\"Int\" Dimensions=\"" + ne + " " + std::to_string(nl) + "\"  Format=\"HDF\">\n" + h5name + ":/CONNECTIVITY\n</DataStructure>\n</Topology>\n");
    P("<Geometry Type=\"X_Y_Z\">\n");
    for (int d = 1; d <= 3; d++)
      P("<DataStructure DataType=\"Double\" Precision=\"8\" Dimensions=\"" + nvt + "  1\"  Format=\"HDF\">\n" + h5name + ":/NODES_X" + std::to_string(d) + "\n</DataStructure>\n");
    P("</Geometry>\n");
    P("<Attribute Name=\"Domain_partitions\" AttributeType=\"Scalar\" Center=\"Cell\">\n<DataItem DataType=\"Double\" Dimensions=\"" + ne + "  1\"  Format=\"HDF\">\n" + h5name +
      ":/DOMAIN_PARTITIONS\n</DataItem>\n</Attribute>\n");
    for (int k = 0; k < nfields; k++)
      P(std::string("<Attribute Name=\"") + names[k] + "\" AttributeType=\"Scalar\" Center=\"Node\">\n<DataItem DataType=\"Double\" Precision=\"8\" Dimensions=\"" + nvt +
        "  1\"  Format=\"HDF\">\n" + h5name + ":/" + names[k] + "\n</DataItem>\n</Attribute>\n");
    P("</Grid>\n</Domain>\n</Xdmf>\n");
    ok = (fclose(f) == 0) && ok;
    FH_REQUIRE(ok, "fh_write_xdmf: write error on %s", xmf.c_str());
  }
  // ---- heavy data ----
  const hid_t file = h.H5Fcreate(h5path.c_str(), 2u /* H5F_ACC_TRUNC */, 0, 0);
  FH_REQUIRE(file >= 0, "fh_write_xdmf: cannot create %s", h5path.c_str());
  int bad = 0;
  std::vector<double> col(std::max(nnode, nel));
  for (int d = 0; d < 3; d++) {
    for (int i = 0; i < nnode; i++) col[i] = d < dim ? xy[(size_t)i * dim + d] : 0.0;
    bad |= put(h, file, (std::string("/NODES_X") + std::to_string(d + 1)).c_str(), *h.native_double, (hsize_t)nnode, col.data());
  }
  {
    std::vector<int> order(nl);                    // Writer_one_level::FemusToVTKorToXDMFConn: the four side-face centres of HEX27 are permuted
    for (int i = 0; i < nl; i++) order[i] = i;
    if (geom == GEOM_HEX) {
      const int perm[27] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 23, 21, 20, 22, 24, 25, 26};
      for (int i = 0; i < 27; i++) order[i] = perm[i];
    }
    std::vector<int> conn((size_t)nel * nl);
    for (int e = 0; e < nel; e++)
      for (int k = 0; k < nl; k++) conn[(size_t)e * nl + k] = ed[(size_t)e * nl + order[k]];
    bad |= put(h, file, "/CONNECTIVITY", *h.native_int, (hsize_t)nel * nl, conn.data());
  }
  std::fill(col.begin(), col.end(), 0.0);         // one rank: subdomain 0 everywhere
  bad |= put(h, file, "/DOMAIN_PARTITIONS", *h.native_double, (hsize_t)nel, col.data());
  const int nv = nvert_of(geom);
  for (int k = 0; k < nfields; k++) {
    const double* v = values[k];
    std::vector<double> full;
    if (fe[k] == 0) {                              // a Q1 field at every biquadratic node: the element interpolation of its vertex values
      full.assign(nnode, 0.0);
      for (int e = 0; e < nel; e++)
        for (int i = 0; i < nl; i++) {
          double s = 0.0;
          int cnt = 0;
          for (int q = 0; q < nv; q++) {
            bool on = true;
            for (int d = 0; d < dim; d++) on = on && (xc(geom, i, d) == 0 || xc(geom, i, d) == xc(geom, q, d));
            if (on) {
              s += v[ed[(size_t)e * nl + q]];
              cnt++;
            }
          }
          full[ed[(size_t)e * nl + i]] = s / cnt;
        }
      v = full.data();
    }
    bad |= put(h, file, (std::string("/") + names[k]).c_str(), *h.native_double, (hsize_t)nnode, v);
  }
  bad |= h.H5Fclose(file) < 0;
  FH_REQUIRE(!bad, "fh_write_xdmf: HDF5 reported an error while writing %s", h5path.c_str());
  return 0;
  FH_GUARD_END("fh_write_xdmf")
}
