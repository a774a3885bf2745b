// TEST INFRASTRUCTURE ONLY (oracle side).  Thin extern "C" accessors over the
// reference's own FE basis / quadrature classes, compiled from the sources
// where they lie under /root/reference (see oracle/Makefile, target _ref).
// Nothing from the reference is copied: this file only *calls* its classes.
//   femus::Gauss            src/02_reference_geom_elements/02_quadrature/quadrature_interface.hpp:32
//   femus::HexBiquadratic   src/02_reference_geom_elements/01_fe/3d/Hexahedron.hpp
//   femus::QuadBiquadratic  src/02_reference_geom_elements/01_fe/2d/Quadrilateral.hpp
#include "quadrature_interface.hpp"
#include "Hexahedron.hpp"
#include "Quadrilateral.hpp"
#include "Edge.hpp"
#include <cstring>
#include <memory>

using namespace femus;

static basis* make_basis(const char* geom, const char* fe) {
  if (!strcmp(geom, "hex")) {
    if (!strcmp(fe, "linear")) return new HexLinear();
    if (!strcmp(fe, "quadratic")) return new HexQuadratic();
    if (!strcmp(fe, "biquadratic")) return new HexBiquadratic();
  } else if (!strcmp(geom, "quad")) {
    if (!strcmp(fe, "linear")) return new QuadLinear();
    if (!strcmp(fe, "quadratic")) return new QuadQuadratic();
    if (!strcmp(fe, "biquadratic")) return new QuadBiquadratic();
  } else if (!strcmp(geom, "line")) {
    if (!strcmp(fe, "linear")) return new LineLinear();
    if (!strcmp(fe, "biquadratic")) return new LineBiquadratic();
  }
  return nullptr;
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
  std::unique_ptr<basis> b(make_basis(geom, fe));
  return b ? b->n_dofs() : -1;
}
int ref_ndofs_fine(const char* geom, const char* fe) {
  std::unique_ptr<basis> b(make_basis(geom, fe));
  return b ? b->n_dofs_fine() : -1;
}

// which: 0 phi, 1 dx, 2 dy, 3 dz, 4 dxx, 5 dyy, 6 dzz, 7 dxy, 8 dyz, 9 dzx ; evaluated for dof j at point pt
double ref_eval(const char* geom, const char* fe, int which, int j, const double* pt) {
  std::unique_ptr<basis> b(make_basis(geom, fe));
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
  std::unique_ptr<basis> b(make_basis(geom, fe));
  for (int d = 0; d < dim; d++) out[d] = b->GetXcoarse(i)[d];
}
void ref_ind(const char* geom, const char* fe, int i, int dim, int* out) {
  std::unique_ptr<basis> b(make_basis(geom, fe));
  for (int d = 0; d < dim; d++) out[d] = b->GetIND(i)[d];
}
void ref_kvert_ind(const char* geom, const char* fe, int i, int* out) {
  std::unique_ptr<basis> b(make_basis(geom, fe));
  out[0] = b->GetKVERT_IND(i)[0];
  out[1] = b->GetKVERT_IND(i)[1];
}
unsigned ref_fine2coarse_vertex(const char* geom, const char* fe, int child, unsigned v) {
  std::unique_ptr<basis> b(make_basis(geom, fe));
  return b->GetFine2CoarseVertexMapping(child, v);
}
unsigned ref_face_dof(const char* geom, const char* fe, unsigned face, unsigned j) {
  std::unique_ptr<basis> b(make_basis(geom, fe));
  return b->GetFaceDof(face, j);
}

}  // extern "C"
