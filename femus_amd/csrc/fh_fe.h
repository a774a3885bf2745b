// host-side reference-element helpers (see fh_fe.cpp)
#pragma once
#include <vector>

namespace fhfe {
enum { GEOM_HEX = 0, GEOM_QUAD = 1, GEOM_LINE = 2, GEOM_TRI = 3, GEOM_TET = 4, GEOM_WEDGE = 5 };
enum { FE_LINEAR = 0, FE_SERENDIPITY = 1, FE_BIQUADRATIC = 2, FE_CONSTANT = 3 };      // FEFamily order of the reference (CONTINUOUS_LINEAR, _SERENDIPITY, _BIQUADRATIC, DISCONTINUOUS_CONSTANT; 4 = DISCONTINUOUS_LINEAR lives in fh_mesh.cpp / fh_ns.hip)
inline bool fe_known(int fe) { return fe >= 0 && fe <= 3; }
int dim_of(int geom);
int nloc_of(int geom);        // biquadratic nodes per element (27 / 9)
int nvert_of(int geom);       // 8 / 4 (= number of children)
int nedge_end_of(int geom);   // end of the edge-node range (20 / 8)
int nfaces_of(int geom);
int ndofs_of(int geom, int fe);
int xc(int geom, int node, int d);   // local node coordinates in {-1,0,1} (tensor-product elements)
void node_ref(int geom, int node, double* pt);   // reference coordinates of a local node, any element (triangle: 0, 1/2, 1, 1/3)
int gauss_npoints(int geom, int order);
int gauss_table(int geom, int order, double* w, double* x);
void eval_basis(int geom, int fe, const double* pt, double* phi, double* dphi);
void eval_basis_d2(int geom, int fe, const double* pt, double* d2phi /* [nc][3 or 6] */);
int shape_tables(int geom, int fe, int order, std::vector<double>& w, std::vector<double>& phi, std::vector<double>& dphi);
void child_node_ref(int geom, int child, int node, double* pt);
int fine2coarse_vertex(int geom, int child, int v);
void elem_prolongator(int geom, int fe, std::vector<double>& P);
// local element nodes of face f in the face element's own node order (QUAD9 / EDGE3 parametrised by the two (one) free
// coordinates in cyclic order); returns the number of face nodes for the FE family
int face_nodes(int geom, int fe, int face, int* out);
}  // namespace fhfe
