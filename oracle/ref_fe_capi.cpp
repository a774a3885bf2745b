// TEST INFRASTRUCTURE ONLY (oracle side).  Thin extern "C" accessors over the
// reference's own FE basis / quadrature classes, compiled from the sources
// where they lie under /root/reference (see oracle/Makefile, target _ref).
// Nothing from the reference is copied: this file only *calls* its classes.
//   femus::Gauss            src/02_reference_geom_elements/02_quadrature/quadrature_interface.hpp:32
//   femus::HexBiquadratic   src/02_reference_geom_elements/01_fe/3d/Hexahedron.hpp
//   femus::QuadBiquadratic  src/02_reference_geom_elements/01_fe/2d/Quadrilateral.hpp
//   femus::TriLinear / TriQuadratic / TriBiquadratic  src/02_reference_geom_elements/01_fe/2d/Triangle.hpp (round 6)
//   femus::HexQuadratic / QuadQuadratic (serendipity), hex0 / quad0 (piecewise constant), hexpwLinear / quadpwLinear: same headers
//   femus::GeomElemBase     src/02_reference_geom_elements/00_definition/GeomElemBase.hpp:32 (build), :83 (get_nodes_of_face), :97 (get_embedding_matrix)
#include "quadrature_interface.hpp"
#include "Hexahedron.hpp"
#include "Quadrilateral.hpp"
#include "Triangle.hpp"
#include "Tetrahedron.hpp"
#include "Wedge.hpp"
#include "Edge.hpp"
#include "GeomElemBase.hpp"
#include <cstring>
#include <map>
#include <memory>
#include <string>

using namespace femus;

static basis* make_basis(const char* geom, const char* fe) {
  if (!strcmp(geom, "hex")) {
    if (!strcmp(fe, "linear")) return new HexLinear();
    if (!strcmp(fe, "quadratic")) return new HexQuadratic();
    if (!strcmp(fe, "biquadratic")) return new HexBiquadratic();
    if (!strcmp(fe, "constant")) return new hex0();
    if (!strcmp(fe, "pwlinear")) return new hexpwLinear();
  } else if (!strcmp(geom, "quad")) {
    if (!strcmp(fe, "linear")) return new QuadLinear();
    if (!strcmp(fe, "quadratic")) return new QuadQuadratic();
    if (!strcmp(fe, "biquadratic")) return new QuadBiquadratic();
    if (!strcmp(fe, "constant")) return new quad0();
    if (!strcmp(fe, "pwlinear")) return new quadpwLinear();
  } else if (!strcmp(geom, "wedge")) {
    if (!strcmp(fe, "linear")) return new WedgeLinear();
    if (!strcmp(fe, "quadratic")) return new WedgeQuadratic();
    if (!strcmp(fe, "biquadratic")) return new WedgeBiquadratic();
  } else if (!strcmp(geom, "tet")) {
    if (!strcmp(fe, "linear")) return new TetLinear();
    if (!strcmp(fe, "quadratic")) return new TetQuadratic();
    if (!strcmp(fe, "biquadratic")) return new TetBiquadratic();
  } else if (!strcmp(geom, "tri")) {
    if (!strcmp(fe, "linear")) return new TriLinear();
    if (!strcmp(fe, "quadratic")) return new TriQuadratic();
    if (!strcmp(fe, "biquadratic")) return new TriBiquadratic();
  } else if (!strcmp(geom, "line")) {
    if (!strcmp(fe, "linear")) return new LineLinear();
    if (!strcmp(fe, "biquadratic")) return new LineBiquadratic();
  }
  return nullptr;
}

// created once per (geom, fe) and kept: the reference's `basis` has no virtual destructor, so nothing is deleted through that type
static basis* cached_basis(const char* geom, const char* fe) {
  static std::map<std::string, basis*> cache;
  const std::string key = std::string(geom) + "/" + fe;
  auto it = cache.find(key);
  if (it == cache.end()) it = cache.emplace(key, make_basis(geom, fe)).first;
  return it->second;
}

extern "C" {

// number of Gauss points; fills w[ng] and x[dim*ng] (x[d*ng+ig]) exactly as the reference tables hold them
int ref_gauss(const char* geom, const char* order, int dim, double* w, double* x) {
  Gauss g(geom, order);
  const int ng = g.GetGaussPointsNumber();
  if (w) for (int i = 0; i < ng; i++) w[i] = g.GetGaussWeightsPointer()[i];
  if (x) for (int d = 0; d < dim; d++) for (int i = 0; i < ng; i++) x[d * ng + i] = g.GetGaussCoordinatePointer(d)[i];
  return ng;
}

int ref_ndofs(const char* geom, const char* fe) {
  basis* b = cached_basis(geom, fe);
  return b ? b->n_dofs() : -1;
}
int ref_ndofs_fine(const char* geom, const char* fe) {
  basis* b = cached_basis(geom, fe);
  return b ? b->n_dofs_fine() : -1;
}

// which: 0 phi, 1 dx, 2 dy, 3 dz, 4 dxx, 5 dyy, 6 dzz, 7 dxy, 8 dyz, 9 dzx ; evaluated for dof j at point pt
double ref_eval(const char* geom, const char* fe, int which, int j, const double* pt) {
  basis* b = cached_basis(geom, fe);
  const int* I = b->GetIND(j);
  switch (which) {
    case 0: return b->eval_phi(I, pt);
    case 1: return b->eval_dphidx(I, pt);
    case 2: return b->eval_dphidy(I, pt);
    case 3: return b->eval_dphidz(I, pt);
    case 4: return b->eval_d2phidx2(I, pt);
    case 5: return b->eval_d2phidy2(I, pt);
    case 6: return b->eval_d2phidz2(I, pt);
    case 7: return b->eval_d2phidxdy(I, pt);
    case 8: return b->eval_d2phidydz(I, pt);
    case 9: return b->eval_d2phidzdx(I, pt);
  }
  return 0.;
}

// topology tables of the Lagrange families (coarse node coords, IND, KVERT_IND, fine2coarse vertex map, face dofs)
void ref_xcoarse(const char* geom, const char* fe, int i, int dim, double* out) {
  basis* b = cached_basis(geom, fe);
  for (int d = 0; d < dim; d++) out[d] = b->GetXcoarse(i)[d];
}
void ref_ind(const char* geom, const char* fe, int i, int dim, int* out) {
  basis* b = cached_basis(geom, fe);
  for (int d = 0; d < dim; d++) out[d] = b->GetIND(i)[d];
}
void ref_kvert_ind(const char* geom, const char* fe, int i, int* out) {
  basis* b = cached_basis(geom, fe);
  out[0] = b->GetKVERT_IND(i)[0];
  out[1] = b->GetKVERT_IND(i)[1];
}
unsigned ref_fine2coarse_vertex(const char* geom, const char* fe, int child, unsigned v) {
  basis* b = cached_basis(geom, fe);
  return b->GetFine2CoarseVertexMapping(child, v);
}
unsigned ref_face_dof(const char* geom, const char* fe, unsigned face, unsigned j) {
  basis* b = cached_basis(geom, fe);
  return b->GetFaceDof(face, j);
}

// fine-node reference coordinates of the family (basis::GetX, Basis.hpp:251): the points set_prolongation_OneElement_All_FE
// (ElemType.cpp:439-532) evaluates the coarse shape functions at
void ref_xfine(const char* geom, const char* fe, int i, int dim, double* out) {
  basis* b = cached_basis(geom, fe);
  for (int d = 0; d < dim; d++) out[d] = b->GetX(i)[d];
}

// GeomElem* topology tables (00_definition): sizes, nodes of a face, the (deprecated) float embedding matrix of a child
int ref_geomelem_info(const char* geom, unsigned fe_family, int* dim, int* n_nodes, int* n_nodes_linear, int* n_faces) {
  std::unique_ptr<GeomElemBase> g = GeomElemBase::build(geom, fe_family);
  g->set_faceNumber_offsets();
  *dim = g->get_dimension();
  *n_nodes = g->n_nodes();
  *n_nodes_linear = g->n_nodes_linear();
  *n_faces = g->n_faces_total();
  return 0;
}
int ref_geomelem_face_nodes(const char* geom, unsigned fe_family, unsigned f, unsigned* out) {
  std::unique_ptr<GeomElemBase> g = GeomElemBase::build(geom, fe_family);
  std::vector<unsigned> v = g->get_nodes_of_face(f);
  for (size_t k = 0; k < v.size(); k++) out[k] = v[k];
  return (int)v.size();
}
double ref_geomelem_embedding(const char* geom, unsigned fe_family, unsigned child, unsigned i, unsigned j) {
  std::unique_ptr<GeomElemBase> g = GeomElemBase::build(geom, fe_family);
  return (double)g->get_embedding_matrix(child, i, j);
}

}  // extern "C"
