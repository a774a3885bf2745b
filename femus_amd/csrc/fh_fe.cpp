// Reference-element data on the host (a1, a2, a3, a6 of SURVEY 8): Gauss tables, Lagrange bases on
// QUAD9 / HEX27, FE-at-quadrature tables, element prolongator.  Pure setup code: the tables are uploaded
// once to the device by the assembler (fh_assemble.hip).
//   Gauss    : src/02_reference_geom_elements/02_quadrature/quadrature_interface.cpp:36-94, 1d/quadrature_Line.cpp,
//              2d/quadrature_Quadrangle.cpp, 3d/quadrature_Hexahedron.cpp (14-significant-digit literals)
//   bases    : 01_fe/1d/Edge.hpp:72-104, 2d/Quadrilateral.cpp:68-110, 3d/Hexahedron.cpp:95-163; serendipity (QuadQuadratic, Quadrilateral.cpp:113-161;
//              HexQuadratic, Hexahedron.cpp:167-256) and piecewise constant (quad0 / hex0, Quadrilateral.hpp:173-, Hexahedron.hpp:196-)
//   tables   : 03_fe_evaluations_at_quadrature/ElemType.cpp:576-741
//   prolong. : 03_fe_evaluations_at_quadrature/ElemType.cpp:439-532
#include "fh_fe.h"
#include <utility>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace fhfe {

// local node coordinates of the FEMuS HEX27 / QUAD9 ordering: 8 vertices, 12 edge mid-points (bottom ring,
// top ring, vertical), 4 side-face centres (y-, x+, y+, x-), bottom, top, centre.
static const signed char XC_HEX[27][3] = {
    {-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1},
    {0, -1, -1},  {1, 0, -1},  {0, 1, -1}, {-1, 0, -1}, {0, -1, 1},  {1, 0, 1},  {0, 1, 1}, {-1, 0, 1},
    {-1, -1, 0},  {1, -1, 0},  {1, 1, 0},  {-1, 1, 0},  {0, -1, 0},  {1, 0, 0},  {0, 1, 0}, {-1, 0, 0},
    {0, 0, -1},   {0, 0, 1},   {0, 0, 0}};
static const signed char XC_QUAD[9][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}, {0, -1}, {1, 0}, {0, 1}, {-1, 0}, {0, 0}};
// EDGE3 (1d/Edge.cpp:22-30): the two end points, then the middle
static const signed char XC_LINE[3][1] = {{-1}, {1}, {0}};
// TRI7 (2d/Triangle.cpp:27-37): vertices, edge middles, centre; TRI_IND = the (i, j) selectors of the basis polynomials (0, 1, 2 along an edge, 7 the bubble);
// children (Triangle.cpp:48-53): three at the vertices, the fourth the middle triangle {4, 5, 3}; faces (:55-59)
static const double XC_TRI[7][2] = {{0, 0}, {1, 0}, {0, 1}, {0.5, 0}, {0.5, 0.5}, {0, 0.5}, {1. / 3., 1. / 3.}};
static const int TRI_IND[7][2] = {{0, 0}, {2, 0}, {0, 2}, {1, 0}, {1, 1}, {0, 1}, {7, 7}};
static const int TRI_F2C[4][3] = {{0, 3, 5}, {3, 1, 4}, {5, 4, 2}, {4, 5, 3}};
static const int TRI_FACE[3][3] = {{0, 1, 3}, {1, 2, 4}, {2, 0, 5}};
// TET10 (3d/Tetrahedron.cpp:24-100; the four face nodes and the centre of the reference's TET15 are not served): vertices, edge middles; selectors; the eight
// children (four at the vertices, four out of the inner octahedron); faces = (three vertices, three middles)
static const double XC_TET[10][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0.5, 0, 0}, {0.5, 0.5, 0}, {0, 0.5, 0}, {0., 0, 0.5}, {0.5, 0., 0.5}, {0, 0.5, 0.5}};
static const int TET_IND[10][3] = {{0, 0, 0}, {2, 0, 0}, {0, 2, 0}, {0, 0, 2}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}};
static const int TET_F2C[8][4] = {{0, 4, 6, 7}, {4, 1, 5, 8}, {6, 5, 2, 9}, {7, 8, 9, 3}, {5, 6, 4, 7}, {8, 7, 5, 4}, {7, 9, 8, 5}, {9, 5, 7, 6}};
static const int TET_FACE[4][6] = {{0, 2, 1, 6, 5, 4}, {0, 1, 3, 4, 8, 7}, {1, 2, 3, 5, 9, 8}, {2, 0, 3, 6, 7, 9}};

int dim_of(int geom) { return (geom == GEOM_HEX || geom == GEOM_TET) ? 3 : (geom == GEOM_QUAD || geom == GEOM_TRI) ? 2 : 1; }
int nloc_of(int geom) { return geom == GEOM_HEX ? 27 : geom == GEOM_QUAD ? 9 : geom == GEOM_TRI ? 7 : geom == GEOM_TET ? 10 : 3; }
int nvert_of(int geom) { return geom == GEOM_HEX ? 8 : (geom == GEOM_QUAD || geom == GEOM_TET) ? 4 : geom == GEOM_TRI ? 3 : 2; }
int nedge_end_of(int geom) { return geom == GEOM_HEX ? 20 : geom == GEOM_QUAD ? 8 : geom == GEOM_TRI ? 6 : geom == GEOM_TET ? 10 : 2; }
int nfaces_of(int geom) { return geom == GEOM_HEX ? 6 : (geom == GEOM_QUAD || geom == GEOM_TET) ? 4 : geom == GEOM_TRI ? 3 : 2; }
// (on the line the "quadratic" family IS the three-node one: NVE[5] = {2, 3, 3, 1, 2}, GeomElTypeEnum)
int ndofs_of(int geom, int fe) {
  return fe == FE_LINEAR ? nvert_of(geom) : fe == FE_SERENDIPITY ? (geom == GEOM_LINE ? 3 : nedge_end_of(geom)) : fe == FE_CONSTANT ? 1 : nloc_of(geom);
}

int xc(int geom, int node, int d) { return geom == GEOM_HEX ? XC_HEX[node][d] : geom == GEOM_QUAD ? XC_QUAD[node][d] : XC_LINE[node][d]; }
void node_ref(int geom, int node, double* pt) {
  for (int k = 0; k < dim_of(geom); k++) pt[k] = geom == GEOM_TRI ? XC_TRI[node][k] : geom == GEOM_TET ? XC_TET[node][k] : (double)xc(geom, node, k);
}

// ---- Gauss-Legendre in extended precision, then the reference's 14-significant-digit rounding -----------
static void gauss_legendre_ld(int n, long double* x, long double* w) {
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < n; i++) {
    long double z = cosl(pi * (i + 0.75L) / (n + 0.5L));
    long double pp = 1;
    for (int it = 0; it < 100; it++) {
      long double p1 = 1, p2 = 0;
      for (int j = 0; j < n; j++) {
        long double p3 = p2;
        p2 = p1;
        p1 = ((2 * j + 1) * z * p2 - j * p3) / (j + 1);
      }
      pp = n * (z * p1 - p2) / (z * z - 1);
      long double z1 = z;
      z = z1 - p1 / pp;
      if (fabsl(z - z1) < 1e-19L) break;
    }
    x[n - 1 - i] = z;  // ascending
    w[n - 1 - i] = 2 / ((1 - z * z) * pp * pp);
  }
  if (n % 2) x[n / 2] = 0;
}

static double round14(long double v) {
  char buf[64];
  snprintf(buf, sizeof(buf), "%.14Lg", v);
  double r = strtod(buf, nullptr);
  return r + 0.0;
}

// triangle rules (2d/quadrature_Triangle.cpp): symmetric rules with the barycentre, 1 / 4 / 7 / 13 / 19 points; first row the weights (the reference
// triangle has area 1/2), then x, y -- the numbers as the reference's tables hold them (14 significant digits, the last rule 7)
static const int TRI_NG[5] = {1, 4, 7, 13, 19};
static const double TRI_G0[3][1] = {{0.5}, {0.33333333333333}, {0.33333333333333}};
static const double TRI_G1[3][4] = {{-0.28125, 0.26041666666667, 0.26041666666667, 0.26041666666667}, {0.33333333333333, 0.6, 0.2, 0.2}, {0.33333333333333, 0.2, 0.6, 0.2}};
static const double TRI_G2[3][7] = {{0.1125, 0.062969590272414, 0.062969590272414, 0.062969590272414, 0.066197076394253, 0.066197076394253, 0.066197076394253},
                                    {0.33333333333333, 0.79742698535309, 0.10128650732346, 0.10128650732346, 0.05971587178977, 0.47014206410511, 0.47014206410511},
                                    {0.33333333333333, 0.10128650732346, 0.79742698535309, 0.10128650732346, 0.47014206410511, 0.05971587178977, 0.47014206410511}};
static const double TRI_G3[3][13] = {
    {-0.074785022233835, 0.087807628716602, 0.087807628716602, 0.087807628716602, 0.026673617804419, 0.026673617804419, 0.026673617804419, 0.038556880445128,
     0.038556880445128, 0.038556880445128, 0.038556880445128, 0.038556880445128, 0.038556880445128},
    {0.33333333333333, 0.47930806784192, 0.26034596607904, 0.26034596607904, 0.86973979419557, 0.065130102902216, 0.065130102902216, 0.63844418856981,
     0.63844418856981, 0.048690315425316, 0.048690315425316, 0.31286549600488, 0.31286549600488},
    {0.33333333333333, 0.26034596607904, 0.47930806784192, 0.26034596607904, 0.065130102902216, 0.86973979419557, 0.065130102902216, 0.048690315425316,
     0.31286549600488, 0.63844418856981, 0.31286549600488, 0.63844418856981, 0.048690315425316}};
static const double TRI_G4[3][19] = {
    {0.0485679, 0.01566735, 0.01566735, 0.01566735, 0.03891377, 0.03891377, 0.03891377, 0.03982387, 0.03982387, 0.03982387, 0.01278884, 0.01278884, 0.01278884,
     0.02164177, 0.02164177, 0.02164177, 0.02164177, 0.02164177, 0.02164177},
    {0.3333333, 0.02063496, 0.4896825, 0.4896825, 0.1258208, 0.4370896, 0.4370896, 0.6235929, 0.1882035, 0.1882035, 0.910541, 0.04472951, 0.04472951, 0.03683841,
     0.03683841, 0.7411986, 0.7411986, 0.221963, 0.221963},
    {0.3333333, 0.4896825, 0.02063496, 0.4896825, 0.4370896, 0.1258208, 0.4370896, 0.1882035, 0.6235929, 0.1882035, 0.04472951, 0.910541, 0.04472951, 0.7411986,
     0.221963, 0.03683841, 0.221963, 0.03683841, 0.7411986}};
static const double* TRI_G[5] = {TRI_G0[0], TRI_G1[0], TRI_G2[0], TRI_G3[0], TRI_G4[0]};
// tetrahedron rules (3d/quadrature_Tetrahedron.cpp): 1 / 5 / 15 / 31 / 45 points; weights (the reference tetrahedron has volume 1/6), then x, y, z -- the numbers
// as the reference's tables hold them
static const int TET_NG[5] = {1, 5, 15, 31, 45};
static const double TET_G0[4][1] = {
    {0.16666666666667},
    {0.25},
    {0.25},
    {0.25}};
static const double TET_G1[4][5] = {
    {-0.13333333333333, 0.075, 0.075, 0.075, 0.075},
    {0.25, 0.5, 0.16666666666667, 0.16666666666667, 0.16666666666667},
    {0.25, 0.16666666666667, 0.5, 0.16666666666667, 0.16666666666667},
    {0.25, 0.16666666666667, 0.16666666666667, 0.5, 0.16666666666667}};
static const double TET_G2[4][15] = {
    {0.030283678097089, 0.006026785714286, 0.006026785714286, 0.006026785714286, 0.006026785714286, 0.011645249086029, 0.011645249086029, 0.011645249086029, 0.011645249086029, 0.010949141561386,
     0.010949141561386, 0.010949141561386, 0.010949141561386, 0.010949141561386, 0.010949141561386},
    {0.25, 0, 0.33333333333333, 0.33333333333333, 0.33333333333333, 0.72727272727273, 0.090909090909091, 0.090909090909091, 0.090909090909091, 0.43344984642634,
     0.43344984642634, 0.43344984642634, 0.066550153573664, 0.066550153573664, 0.066550153573664},
    {0.25, 0.33333333333333, 0, 0.33333333333333, 0.33333333333333, 0.090909090909091, 0.72727272727273, 0.090909090909091, 0.090909090909091, 0.43344984642634,
     0.066550153573664, 0.066550153573664, 0.43344984642634, 0.43344984642634, 0.066550153573664},
    {0.25, 0.33333333333333, 0.33333333333333, 0, 0.33333333333333, 0.090909090909091, 0.090909090909091, 0.72727272727273, 0.090909090909091, 0.066550153573664,
     0.43344984642634, 0.066550153573664, 0.43344984642634, 0.066550153573664, 0.43344984642634}};
static const double TET_G3[4][31] = {
    {0.01826422, 0.01059994, 0.01059994, 0.01059994, 0.01059994, -0.06251774, -0.06251774, -0.06251774, -0.06251774, 0.004891425,
     0.004891425, 0.004891425, 0.004891425, 0.0009700176, 0.0009700176, 0.0009700176, 0.0009700176, 0.0009700176, 0.0009700176, 0.02755732,
     0.02755732, 0.02755732, 0.02755732, 0.02755732, 0.02755732, 0.02755732, 0.02755732, 0.02755732, 0.02755732, 0.02755732,
     0.02755732},
    {0.25, 0.7653604, 0.07821319, 0.07821319, 0.07821319, 0.6344704, 0.1218432, 0.1218432, 0.1218432, 0.002382507,
     0.3325392, 0.3325392, 0.3325392, 0, 0, 0, 0.5, 0.5, 0.5, 0.6,
     0.6, 0.6, 0.2, 0.2, 0.2, 0.1, 0.1, 0.1, 0.1, 0.1,
     0.1},
    {0.25, 0.07821319, 0.7653604, 0.07821319, 0.07821319, 0.1218432, 0.6344704, 0.1218432, 0.1218432, 0.3325392,
     0.002382507, 0.3325392, 0.3325392, 0, 0.5, 0.5, 0, 0, 0.5, 0.2,
     0.1, 0.1, 0.6, 0.1, 0.1, 0.6, 0.6, 0.2, 0.2, 0.1,
     0.1},
    {0.25, 0.07821319, 0.07821319, 0.7653604, 0.07821319, 0.1218432, 0.1218432, 0.6344704, 0.1218432, 0.3325392,
     0.3325392, 0.002382507, 0.3325392, 0.5, 0, 0.5, 0, 0.5, 0, 0.1,
     0.2, 0.1, 0.1, 0.6, 0.1, 0.2, 0.1, 0.6, 0.1, 0.6,
     0.2}};
static const double TET_G4[4][45] = {
    {-0.03932701, 0.004081316, 0.004081316, 0.004081316, 0.004081316, 0.0006580868, 0.0006580868, 0.0006580868, 0.0006580868, 0.004384259,
     0.004384259, 0.004384259, 0.004384259, 0.004384259, 0.004384259, 0.01383006, 0.01383006, 0.01383006, 0.01383006, 0.01383006,
     0.01383006, 0.004240437, 0.004240437, 0.004240437, 0.004240437, 0.004240437, 0.004240437, 0.004240437, 0.004240437, 0.004240437,
     0.004240437, 0.004240437, 0.004240437, 0.00223874, 0.00223874, 0.00223874, 0.00223874, 0.00223874, 0.00223874, 0.00223874,
     0.00223874, 0.00223874, 0.00223874, 0.00223874, 0.00223874},
    {0.25, 0.6175872, 0.1274709, 0.1274709, 0.1274709, 0.9037635, 0.03207883, 0.03207883, 0.03207883, 0.4502229,
     0.4502229, 0.4502229, 0.0497771, 0.0497771, 0.0497771, 0.3162696, 0.3162696, 0.3162696, 0.1837304, 0.1837304,
     0.1837304, 0.51328, 0.51328, 0.51328, 0.02291779, 0.02291779, 0.02291779, 0.2319011, 0.2319011, 0.2319011,
     0.2319011, 0.2319011, 0.2319011, 0.1937465, 0.1937465, 0.1937465, 0.7303134, 0.7303134, 0.7303134, 0.03797005,
     0.03797005, 0.03797005, 0.03797005, 0.03797005, 0.03797005},
    {0.25, 0.1274709, 0.6175872, 0.1274709, 0.1274709, 0.03207883, 0.9037635, 0.03207883, 0.03207883, 0.4502229,
     0.0497771, 0.0497771, 0.4502229, 0.4502229, 0.0497771, 0.3162696, 0.1837304, 0.1837304, 0.3162696, 0.3162696,
     0.1837304, 0.02291779, 0.2319011, 0.2319011, 0.51328, 0.2319011, 0.2319011, 0.51328, 0.51328, 0.02291779,
     0.02291779, 0.2319011, 0.2319011, 0.7303134, 0.03797005, 0.03797005, 0.1937465, 0.03797005, 0.03797005, 0.1937465,
     0.1937465, 0.7303134, 0.7303134, 0.03797005, 0.03797005},
    {0.25, 0.1274709, 0.1274709, 0.6175872, 0.1274709, 0.03207883, 0.03207883, 0.9037635, 0.03207883, 0.0497771,
     0.4502229, 0.0497771, 0.4502229, 0.0497771, 0.4502229, 0.1837304, 0.3162696, 0.1837304, 0.3162696, 0.1837304,
     0.3162696, 0.2319011, 0.02291779, 0.2319011, 0.2319011, 0.51328, 0.2319011, 0.02291779, 0.2319011, 0.51328,
     0.2319011, 0.51328, 0.02291779, 0.03797005, 0.7303134, 0.03797005, 0.03797005, 0.1937465, 0.03797005, 0.7303134,
     0.03797005, 0.1937465, 0.03797005, 0.1937465, 0.7303134}};
static const double* TET_G[5] = {TET_G0[0], TET_G1[0], TET_G2[0], TET_G3[0], TET_G4[0]};

int gauss_npoints(int geom, int order) {
  if (geom == GEOM_TRI) return TRI_NG[order];
  if (geom == GEOM_TET) return TET_NG[order];
  int n = order + 1, d = (geom == GEOM_LINE) ? 1 : dim_of(geom), r = 1;
  for (int k = 0; k < d; k++) r *= n;
  return r;
}

// w[ng], x[d*ng + ig]; first coordinate slowest, as the reference tables
int gauss_table(int geom, int order, double* w, double* x) {
  if (order < 0 || order > 4) return 1;
  if (geom == GEOM_TET) {
    const int ng = TET_NG[order];
    for (int ig = 0; ig < ng; ig++) {
      if (w) w[ig] = TET_G[order][ig];
      if (x)
        for (int k = 0; k < 3; k++) x[k * ng + ig] = TET_G[order][(k + 1) * ng + ig];
    }
    return 0;
  }
  if (geom == GEOM_TRI) {
    const int ng = TRI_NG[order];
    for (int ig = 0; ig < ng; ig++) {
      if (w) w[ig] = TRI_G[order][ig];
      if (x) {
        x[ig] = TRI_G[order][ng + ig];
        x[ng + ig] = TRI_G[order][2 * ng + ig];
      }
    }
    return 0;
  }
  const int n = order + 1;
  const int d = (geom == GEOM_LINE) ? 1 : dim_of(geom);
  long double x1[8], w1[8];
  gauss_legendre_ld(n, x1, w1);
  const int ng = gauss_npoints(geom, order);
  // the reference's hex "seventh" literals deviate from round14(exact) in the last digit
  // (3d/quadrature_Hexahedron.cpp Gauss3): classes by the number of inner 1-D points among (i,j,k)
  static const double HEX4[4] = {0.042091477490532, 0.078911515795071, 0.14794033605678, 0.27735296695391};
  for (int ig = 0; ig < ng; ig++) {
    int idx[3] = {0, 0, 0};
    int r = ig;
    for (int k = d - 1; k >= 0; k--) {
      idx[k] = r % n;
      r /= n;
    }
    long double ww = 1;
    int inner = 0;
    for (int k = 0; k < d; k++) {
      ww *= w1[idx[k]];
      inner += (idx[k] > 0 && idx[k] < n - 1);
      if (x) x[k * ng + ig] = (n == 1) ? 0.0 : round14(x1[idx[k]]);
    }
    if (w) {
      if (n == 1) w[ig] = (double)ww;
      else if (geom == GEOM_HEX && n == 4) w[ig] = HEX4[inner];
      else w[ig] = round14(ww);
    }
  }
  return 0;
}

// ---- 1-D Lagrange polynomials, same expressions as Edge.hpp:72-104 ------------------------------------
static inline double lagL(double x, int i) { return (!i) * 0.5 * (1. - x) + !(i - 2) * 0.5 * (1. + x); }
static inline double dlagL(double, int i) { return (!i) * (-0.5) + !(i - 2) * 0.5; }
static inline double lagB(double x, int i) { return !i * 0.5 * x * (x - 1.) + !(i - 1) * (1. - x) * (1. + x) + !(i - 2) * 0.5 * x * (1. + x); }
static inline double dlagB(double x, int i) { return !i * (x - 0.5) + !(i - 1) * (-2. * x) + !(i - 2) * (x + 0.5); }

static inline double d2lagB(int i) { return !i * 1.0 + !(i - 1) * (-2.0) + !(i - 2) * 1.0; }
// "quadratic" 1-D factors of the serendipity families (Edge.hpp:81-91): linear at the end nodes, the bubble at the middle one
static inline double lagQ(double x, int i) { return !i * (0.5) * (1. - x) + !(i - 1) * (1. - x) * (1. + x) + !(i - 2) * (0.5) * (1. + x); }
static inline double dlagQ(double x, int i) { return (!i) * (-0.5) + !(i - 1) * (-2. * x) + !(i - 2) * (0.5); }
static inline double d2lagQ(int i) { return !(i - 1) * (-2.); }

// Triangle families (2d/Triangle.hpp:69-181): P1, P2 and P2 enriched with the cubic bubble (TRI7), selected by the (i, j) pair of the node as the 1-D factors
// above are by their index; the terms in the reference's order (the tables are compared bit for bit with the ones its compiled classes give).
// out: phi, d/dx, d/dy, d2/dx2, d2/dy2, d2/dxdy
static void tri_node(int fe, int i, int j, double x, double y, double out[6]) {
  for (int k = 0; k < 6; k++) out[k] = 0.0;
  if (fe == FE_LINEAR) {
    out[0] = (!i * !j) * (1. - x - y) + !(i - 2) * x + !(j - 2) * y;
    out[1] = -(!i * !j) + !(i - 2);
    out[2] = -(!i * !j) + !(j - 2);
  } else if (fe == FE_SERENDIPITY) {
    out[0] = !i * (!j * (1. - x - y) * (1. - 2. * x - 2. * y) + !(j - 1) * 4. * y * (1. - x - y) + !(j - 2) * (-y + 2. * y * y)) +
             !(i - 1) * (!j * 4. * x * (1. - x - y) + !(j - 1) * 4. * x * y) + !(i - 2) * (!j * (-x + 2. * x * x));
    out[1] = !i * (!j * (-3. + 4. * x + 4. * y) + !(j - 1) * y * (-4.)) + !(i - 1) * (!j * 4. * (1. - 2. * x - y) + !(j - 1) * y * (4.)) + !(i - 2) * (!j * (-1 + 4. * x));
    out[2] = !j * (!i * (-3. + 4. * y + 4. * x) + !(i - 1) * x * (-4.)) + !(j - 1) * (!i * 4. * (1. - 2. * y - x) + !(i - 1) * x * (4.)) + !(j - 2) * (!i * (-1 + 4. * y));
    out[3] = !j * ((!i) * 4. + !(i - 1) * (-8.) + !(i - 2) * 4.);
    out[4] = !i * ((!j) * 4. + !(j - 1) * (-8.) + !(j - 2) * 4.);
    out[5] = ((!i) * (!j) + !(i - 1) * !(j - 1)) * 4. + (!(i - 1) * (!j) + (!i) * !(j - 1)) * (-4.);
  } else {
    const double b3 = 3. * x * y * (1 - x - y), bx = y - 2. * x * y - y * y, by = x - x * x - 2. * x * y, bxy = 1 - 2. * x - 2. * y;      // (the products associate as in the reference's inline terms)
    out[0] = !i * (!j * ((1. - x - y) * (1. - 2. * x - 2. * y) + b3) + !(j - 1) * 4. * (y * (1. - x - y) - b3) + !(j - 2) * (-y + 2. * y * y + b3)) +
             !(i - 1) * (!j * 4. * (x * (1. - x - y) - b3) + !(j - 1) * 4. * (x * y - b3)) + !(i - 2) * (!j * (-x + 2. * x * x + b3)) +
             !(i - 7) * (!(j - 7) * 27. * x * y * (1 - x - y));
    out[1] = !i * (!j * (-3. + 4. * x + 4. * y + 3. * bx) + !(j - 1) * 4. * (-y - 3. * bx) + !(j - 2) * 3. * bx) +
             !(i - 1) * (!j * 4. * (1. - 2. * x - y - 3. * bx) + !(j - 1) * 4. * (y - 3. * bx)) + !(i - 2) * (!j * (-1 + 4. * x + 3. * bx)) + !(i - 7) * (!(j - 7) * 27. * bx);
    out[2] = !j * (!i * (-3. + 4. * y + 4. * x + 3. * by) + !(i - 1) * 4. * (-x - 3. * by) + !(i - 2) * 3. * by) +
             !(j - 1) * (!i * 4. * (1. - 2. * y - x - 3. * by) + !(i - 1) * 4. * (x - 3. * by)) + !(j - 2) * (!i * (-1 + 4. * y + 3. * by)) + !(j - 7) * (!(i - 7) * 27. * by);
    out[3] = !i * (!j * (4. - 6. * y) + !(j - 1) * 4. * (6. * y) + !(j - 2) * (-6. * y)) + !(i - 1) * (!j * 4. * (-2. + 6. * y) + !(j - 1) * 4. * (6. * y)) +
             !(i - 2) * (!j * (4. - 6. * y)) + !(i - 7) * (!(j - 7) * (-54. * y));
    out[4] = !j * (!i * (4. - 6. * x) + !(i - 1) * 4. * (6. * x) + !(i - 2) * (-6. * x)) + !(j - 1) * (!i * 4. * (-2. + 6. * x) + !(i - 1) * 4. * (6. * x)) +
             !(j - 2) * (!i * (4. - 6. * x)) + !(j - 7) * (!(i - 7) * (-54. * x));
    out[5] = !j * (!i * (4. + 3. * bxy) + !(i - 1) * 4. * (-1. - 3. * bxy) + !(i - 2) * 3. * bxy) +
             !(j - 1) * (!i * 4. * (-1. - 3. * bxy) + !(i - 1) * 4. * (1. - 3. * bxy)) + !(j - 2) * (!i * (3. * bxy)) + !(j - 7) * (!(i - 7) * 27. * bxy);
  }
}

// Tetrahedron families (3d/Tetrahedron.cpp: TetLinear, TetQuadratic), selected by the (i, j, k) triple of the node; the terms in the reference's order.
// out: phi, d/dx, d/dy, d/dz, then xx, yy, zz, xy, yz, zx (the second derivatives of P2 are the constants 4, -8, -4 of its barycentric products)
static void tet_node(int fe, int i, int j, int k, double x, double y, double z, double out[10]) {
  for (int q = 0; q < 10; q++) out[q] = 0.0;
  if (fe == FE_LINEAR) {
    out[0] = (!i * !j * !k) * (1. - x - y - z) + !(i - 2) * x + !(j - 2) * y + !(k - 2) * z;
    out[1] = -(!i * !j * !k) + !(i - 2);
    out[2] = -(!i * !j * !k) + !(j - 2);
    out[3] = -(!i * !j * !k) + !(k - 2);
    return;
  }
  const double t = 1. - (x + y + z);
  out[0] = !i * (!j * (!k * t * (2. * t - 1.) + !(k - 1) * 4. * z * t + !(k - 2) * (-z + 2. * z * z)) + !(j - 1) * (!k * 4. * y * t + !(k - 1) * 4. * y * z) +
                 !(j - 2) * (!k * (-y + 2. * y * y))) +
           !(i - 1) * (!j * (!k * 4. * x * t + !(k - 1) * 4. * x * z) + !(j - 1) * (!k * 4. * x * y)) + !(i - 2) * (!j * (!k * (-x + 2. * x * x)));
  out[1] = !i * (!j * (!k * (-4. * t + 1.) + !(k - 1) * (-4.) * z) + !(j - 1) * (!k * (-4.) * y)) +
           !(i - 1) * (!j * (!k * 4. * (t - x) + !(k - 1) * 4. * z) + !(j - 1) * (!k * 4. * y)) + !(i - 2) * (!j * (!k * (-1. + 4. * x)));
  out[2] = !i * (!j * (!k * (-4. * t + 1.) + !(k - 1) * (-4.) * z) + !(j - 1) * (!k * 4. * (t - y) + !(k - 1) * 4. * z) + !(j - 2) * (!k * (-1. + 4. * y))) +
           !(i - 1) * (!j * (!k * (-4.) * x) + !(j - 1) * (!k * 4. * x));
  out[3] = !i * (!j * (!k * (-4. * t + 1.) + !(k - 1) * 4. * (t - z) + !(k - 2) * (-1 + 4. * z)) + !(j - 1) * (!k * (-4.) * y + !(k - 1) * 4. * y)) +
           !(i - 1) * (!j * (!k * (-4.) * x + !(k - 1) * 4. * x));
  // Hessian of P2: node (i, j, k) -> barycentric pair; phi = L_a (2 L_a - 1) at a vertex, 4 L_a L_b on an edge; L_0 = t has gradient (-1, -1, -1), L_m the unit vector e_m
  int a = -1, b = -1;                                   // barycentric indices 0 (t), 1 (x), 2 (y), 3 (z) of the node's one or two factors
  const int idx[3] = {i, j, k};
  for (int m = 0; m < 3; m++)
    if (idx[m] == 2) a = b = m + 1;
  if (a < 0) {
    for (int m = 0; m < 3; m++)
      if (idx[m] == 1) (a < 0 ? a : b) = m + 1;
    if (a < 0) a = b = 0;                               // (0, 0, 0): the vertex at the origin
    else if (b < 0) b = 0;                              // one index 1: the edge towards the origin
  }
  auto g = [](int L, int d) { return L == 0 ? -1.0 : (L == d + 1 ? 1.0 : 0.0); };
  const int pr[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {1, 2}, {2, 0}};
  for (int q = 0; q < 6; q++) {
    const int p = pr[q][0], r = pr[q][1];
    out[4 + q] = (a == b) ? 4.0 * g(a, p) * g(a, r) : 4.0 * (g(a, p) * g(b, r) + g(b, p) * g(a, r));
  }
}

// Serendipity bases, the expressions of QuadQuadratic / HexQuadratic term by term and in their order (the tables are compared bit for bit with the ones the
// reference's compiled classes give): a vertex function is the product of the three (two) linear factors times (-2 + ix x + jx y + kx z) ((-1 + ...) in 2-D),
// an edge function the plain product.  out: phi, d/dx, d/dy, d/dz, then xx, yy, zz, xy, yz, zx (2-D: phi, dx, dy, -, xx, yy, -, xy)
static void serendipity_node(int geom, int j, const double* x, double out[10]) {
  const int d = dim_of(geom);
  int I[3] = {1, 1, 1};
  for (int k = 0; k < d; k++) I[k] = xc(geom, j, k) + 1;
  for (int k = 0; k < 10; k++) out[k] = 0.0;
  if (d == 2) {
    const double ix = I[0] - 1., jx = I[1] - 1.;
    const double l0 = lagQ(x[0], I[0]), l1 = lagQ(x[1], I[1]), d0 = dlagQ(x[0], I[0]), d1 = dlagQ(x[1], I[1]), s0 = d2lagQ(I[0]), s1 = d2lagQ(I[1]);
    if (fabs(ix * jx) == 0) {
      out[0] = l0 * l1;
      out[1] = d0 * l1;
      out[2] = l0 * d1;
      out[4] = s0 * l1;
      out[5] = l0 * s1;
      out[7] = d0 * d1;
    } else {
      const double s = -1. + ix * x[0] + jx * x[1];
      out[0] = s * l0 * l1;
      out[1] = l1 * (ix * l0 + s * d0);
      out[2] = l0 * (jx * l1 + s * d1);
      out[4] = l1 * (2. * ix * d0 + s * s0);
      out[5] = l0 * (2. * jx * d1 + s * s1);
      out[7] = ix * l0 * d1 + jx * l1 * d0 + s * d0 * d1;
    }
    return;
  }
  const double ix = I[0] - 1., jx = I[1] - 1., kx = I[2] - 1.;
  const double l0 = lagQ(x[0], I[0]), l1 = lagQ(x[1], I[1]), l2 = lagQ(x[2], I[2]);
  const double d0 = dlagQ(x[0], I[0]), d1 = dlagQ(x[1], I[1]), d2 = dlagQ(x[2], I[2]);
  const double s0 = d2lagQ(I[0]), s1 = d2lagQ(I[1]), s2 = d2lagQ(I[2]);
  if (fabs(ix * jx * kx) == 0) {
    out[0] = l0 * l1 * l2;
    out[1] = d0 * l1 * l2;
    out[2] = l0 * d1 * l2;
    out[3] = l0 * l1 * d2;
    out[4] = s0 * l1 * l2;
    out[5] = l0 * s1 * l2;
    out[6] = l0 * l1 * s2;
    out[7] = d0 * d1 * l2;
    out[8] = l0 * d1 * d2;
    out[9] = d0 * l1 * d2;
  } else {
    const double s = -2. + ix * x[0] + jx * x[1] + kx * x[2];
    out[0] = s * l0 * l1 * l2;
    out[1] = l1 * l2 * (ix * l0 + s * d0);
    out[2] = l0 * l2 * (jx * l1 + s * d1);
    out[3] = l0 * l1 * (kx * l2 + s * d2);
    out[4] = l1 * l2 * (2. * ix * d0 + s * s0);
    out[5] = l2 * l0 * (2. * jx * d1 + s * s1);
    out[6] = l0 * l1 * (2. * kx * d2 + s * s2);
    out[7] = l2 * (ix * l0 * d1 + jx * l1 * d0 + s * d0 * d1);
    out[8] = l0 * (jx * l1 * d2 + kx * l2 * d1 + s * d1 * d2);
    out[9] = l1 * (kx * l2 * d0 + ix * l0 * d2 + s * d2 * d0);
  }
}

// second derivatives, node-major [nc][nh]: 3-D (xx, yy, zz, xy, yz, zx), 2-D (xx, yy, xy) -- the order of elem_type's _d2phidxi2, _d2phideta2,
// _d2phidzeta2, _d2phidxideta, _d2phidetadzeta, _d2phidzetadxi (ElemType.cpp:637-741).  The pure second derivatives of the (bi/tri)linear
// family are identically zero; the mixed ones are not.
void eval_basis_d2(int geom, int fe, const double* pt, double* d2phi) {
  const int d = dim_of(geom), nc = ndofs_of(geom, fe);
  if (fe == FE_CONSTANT) {
    for (int k = 0; k < (d == 1 ? 1 : d == 2 ? 3 : 6); k++) d2phi[k] = 0.0;
    return;
  }
  if (geom == GEOM_TET) {        // (xx, yy, zz, xy, yz, zx)
    for (int j = 0; j < nc; j++) {
      double v[10];
      tet_node(fe, TET_IND[j][0], TET_IND[j][1], TET_IND[j][2], pt[0], pt[1], pt[2], v);
      for (int q = 0; q < 6; q++) d2phi[j * 6 + q] = v[4 + q];
    }
    return;
  }
  if (geom == GEOM_TRI) {        // (xx, yy, xy)
    for (int j = 0; j < nc; j++) {
      double v[6];
      tri_node(fe, TRI_IND[j][0], TRI_IND[j][1], pt[0], pt[1], v);
      d2phi[j * 3 + 0] = v[3];
      d2phi[j * 3 + 1] = v[4];
      d2phi[j * 3 + 2] = v[5];
    }
    return;
  }
  if (fe == FE_SERENDIPITY && d > 1) {
    for (int j = 0; j < nc; j++) {
      double v[10];
      serendipity_node(geom, j, pt, v);
      if (d == 2) {
        d2phi[j * 3 + 0] = v[4]; d2phi[j * 3 + 1] = v[5]; d2phi[j * 3 + 2] = v[7];
      } else {
        for (int k = 0; k < 6; k++) d2phi[j * 6 + k] = v[4 + k];
      }
    }
    return;
  }
  for (int j = 0; j < nc; j++) {
    double l[3], dl[3], d2l[3];
    for (int k = 0; k < d; k++) {
      const int I = xc(geom, j, k) + 1;
      l[k] = (fe == FE_LINEAR) ? lagL(pt[k], I) : lagB(pt[k], I);
      dl[k] = (fe == FE_LINEAR) ? dlagL(pt[k], I) : dlagB(pt[k], I);
      d2l[k] = (fe == FE_LINEAR) ? 0.0 : d2lagB(I);
    }
    if (d == 1) {
      d2phi[j] = d2l[0];
    } else if (d == 2) {
      d2phi[j * 3 + 0] = d2l[0] * l[1];
      d2phi[j * 3 + 1] = l[0] * d2l[1];
      d2phi[j * 3 + 2] = dl[0] * dl[1];
    } else {
      d2phi[j * 6 + 0] = d2l[0] * l[1] * l[2];
      d2phi[j * 6 + 1] = l[0] * d2l[1] * l[2];
      d2phi[j * 6 + 2] = l[0] * l[1] * d2l[2];
      d2phi[j * 6 + 3] = dl[0] * dl[1] * l[2];
      d2phi[j * 6 + 4] = l[0] * dl[1] * dl[2];
      d2phi[j * 6 + 5] = dl[0] * l[1] * dl[2];
    }
  }
}

void eval_basis(int geom, int fe, const double* pt, double* phi, double* dphi /* [nc*dim] node-major */) {
  const int d = dim_of(geom), nc = ndofs_of(geom, fe);
  if (geom == GEOM_TET && fe != FE_CONSTANT) {
    for (int j = 0; j < nc; j++) {
      double v[10];
      tet_node(fe, TET_IND[j][0], TET_IND[j][1], TET_IND[j][2], pt[0], pt[1], pt[2], v);
      if (phi) phi[j] = v[0];
      if (dphi)
        for (int q = 0; q < 3; q++) dphi[j * 3 + q] = v[1 + q];
    }
    return;
  }
  if (geom == GEOM_TRI && fe != FE_CONSTANT) {
    for (int j = 0; j < nc; j++) {
      double v[6];
      tri_node(fe, TRI_IND[j][0], TRI_IND[j][1], pt[0], pt[1], v);
      if (phi) phi[j] = v[0];
      if (dphi) {
        dphi[j * 2 + 0] = v[1];
        dphi[j * 2 + 1] = v[2];
      }
    }
    return;
  }
  if (fe == FE_CONSTANT) {        // quad0 / hex0: the constant one
    if (phi) phi[0] = 1.;
    if (dphi)
      for (int k = 0; k < d; k++) dphi[k] = 0.;
    return;
  }
  if (fe == FE_SERENDIPITY && d > 1) {
    for (int j = 0; j < nc; j++) {
      double v[10];
      serendipity_node(geom, j, pt, v);
      if (phi) phi[j] = v[0];
      if (dphi)
        for (int k = 0; k < d; k++) dphi[j * d + k] = v[1 + k];
    }
    return;
  }
  for (int j = 0; j < nc; j++) {
    double l[3], dl[3];
    for (int k = 0; k < d; k++) {
      const int I = xc(geom, j, k) + 1;
      l[k] = (fe == FE_LINEAR) ? lagL(pt[k], I) : lagB(pt[k], I);
      dl[k] = (fe == FE_LINEAR) ? dlagL(pt[k], I) : dlagB(pt[k], I);
    }
    if (d == 1) {
      if (phi) phi[j] = l[0];
      if (dphi) dphi[j] = dl[0];
    } else if (d == 2) {
      if (phi) phi[j] = l[0] * l[1];
      if (dphi) {
        dphi[j * 2 + 0] = dl[0] * l[1];
        dphi[j * 2 + 1] = l[0] * dl[1];
      }
    } else {
      if (phi) phi[j] = l[0] * l[1] * l[2];
      if (dphi) {
        dphi[j * 3 + 0] = dl[0] * l[1] * l[2];
        dphi[j * 3 + 1] = l[0] * dl[1] * l[2];
        dphi[j * 3 + 2] = l[0] * l[1] * dl[2];
      }
    }
  }
}

int shape_tables(int geom, int fe, int order, std::vector<double>& w, std::vector<double>& phi, std::vector<double>& dphi) {
  const int d = dim_of(geom), nc = ndofs_of(geom, fe), ng = gauss_npoints(geom, order);
  w.resize(ng);
  std::vector<double> x((size_t)d * ng);
  if (gauss_table(geom, order, w.data(), x.data())) return 1;
  phi.resize((size_t)ng * nc);
  dphi.resize((size_t)ng * nc * d);   // [ig][node][dim]
  for (int ig = 0; ig < ng; ig++) {
    double pt[3] = {0, 0, 0};
    for (int k = 0; k < d; k++) pt[k] = x[k * ng + ig];
    eval_basis(geom, fe, pt, &phi[(size_t)ig * nc], &dphi[(size_t)ig * nc * d]);
  }
  return 0;
}

// child j = sub-element at coarse vertex j; local node i of child j sits at (Xc[j] + Xc[i]) / 2
void child_node_ref(int geom, int child, int node, double* pt) {
  if (geom == GEOM_TET) {       // the child's reference tetrahedron mapped affinely onto its four vertices in the father
    const double* v0 = XC_TET[TET_F2C[child][0]];
    for (int k = 0; k < 3; k++) {
      pt[k] = v0[k];
      for (int m = 0; m < 3; m++) pt[k] += (XC_TET[TET_F2C[child][m + 1]][k] - v0[k]) * XC_TET[node][m];
    }
    return;
  }
  if (geom == GEOM_TRI) {       // the child's reference triangle mapped affinely onto its three vertices in the father (the fourth child is the rotated middle one)
    const double* v0 = XC_TRI[TRI_F2C[child][0]];
    const double* v1 = XC_TRI[TRI_F2C[child][1]];
    const double* v2 = XC_TRI[TRI_F2C[child][2]];
    for (int k = 0; k < 2; k++) pt[k] = v0[k] + (v1[k] - v0[k]) * XC_TRI[node][0] + (v2[k] - v0[k]) * XC_TRI[node][1];
    return;
  }
  for (int k = 0; k < dim_of(geom); k++) pt[k] = 0.5 * (xc(geom, child, k) + xc(geom, node, k));
}

int fine2coarse_vertex(int geom, int child, int v) {
  if (geom == GEOM_TRI) return TRI_F2C[child][v];
  if (geom == GEOM_TET) return TET_F2C[child][v];
  double pt[3];
  child_node_ref(geom, child, v, pt);
  for (int n = 0; n < nloc_of(geom); n++) {
    bool same = true;
    for (int k = 0; k < dim_of(geom); k++) same &= (xc(geom, n, k) == pt[k]);
    if (same) return n;
  }
  return -1;
}

void elem_prolongator(int geom, int fe, std::vector<double>& P) {
  const int nch = geom == GEOM_TRI ? 4 : geom == GEOM_TET ? 8 : nvert_of(geom), nc = ndofs_of(geom, fe);
  P.assign((size_t)nch * nc * nc, 0.0);
  std::vector<double> phi(nc);
  for (int j = 0; j < nch; j++)
    for (int i = 0; i < nc; i++) {
      double pt[3];
      child_node_ref(geom, j, i, pt);
      eval_basis(geom, fe, pt, phi.data(), nullptr);
      for (int k = 0; k < nc; k++) P[((size_t)j * nc + i) * nc + k] = (fabs(phi[k]) >= 1.0e-14) ? phi[k] : 0.0;
    }
}

int face_nodes(int geom, int fe, int face, int* out) {
  if (geom == GEOM_TET) {         // triangles: the three vertices, then the three middles (tet_lag faceDofs; TRI6 order)
    const int n = fe == FE_CONSTANT ? 0 : fe == FE_LINEAR ? 3 : 6;
    for (int k = 0; k < n; k++) out[k] = TET_FACE[face][k];
    return n;
  }
  if (geom == GEOM_TRI) {         // edges: the two ends, then the middle (tri_lag faceDofs)
    const int n = fe == FE_CONSTANT ? 0 : fe == FE_LINEAR ? 2 : 3;
    for (int k = 0; k < n; k++) out[k] = TRI_FACE[face][k];
    return n;
  }
  if (geom == GEOM_LINE) {        // the "faces" of a line element are its end points (line_lag faceDofs)
    if (fe == FE_CONSTANT) return 0;
    out[0] = face;
    return 1;
  }
  const int d = dim_of(geom);
  const int centre = (geom == GEOM_HEX) ? 20 + face : 4 + face;
  int d0 = 0;
  for (int k = 0; k < d; k++)
    if (xc(geom, centre, k) != 0) d0 = k;
  const int sgn = xc(geom, centre, d0);
  const int fgeom = (geom == GEOM_HEX) ? GEOM_QUAD : GEOM_LINE;
  const int nfn_q2 = (geom == GEOM_HEX) ? 9 : 3, nfn_q1 = (geom == GEOM_HEX) ? 4 : 2, nfn_ser = (geom == GEOM_HEX) ? 8 : 3;
  const int nfn = (fe == FE_LINEAR) ? nfn_q1 : (fe == FE_SERENDIPITY) ? nfn_ser : (fe == FE_CONSTANT) ? 0 : nfn_q2;
  // free coordinates in cyclic order after d0, oriented so that the normal elem_type::JacobianSur derives from the node order (t_a x t_b on a
  // quadrilateral face, (t_y, -t_x) on an edge) points OUT of the element, as with the reference's own face tables (hex_lag / quad_lag faceDofs;
  // the sign matters to vector-valued boundary terms such as the pressure integral of 03_navier_stokes.hpp:185-290)
  int a = (d0 + 1) % d, b = (d0 + 2) % d;
  bool reverse = false;
  if (d == 3) {
    if (sgn < 0) std::swap(a, b);
  } else {
    reverse = (d0 == 0) ? (sgn < 0) : (sgn > 0);
  }
  for (int i = 0; i < nfn; i++) {
    int xi, eta = 0;
    if (fgeom == GEOM_QUAD) {
      xi = XC_QUAD[i][0];
      eta = XC_QUAD[i][1];
    } else {
      static const int XC_LINE[3] = {-1, 1, 0};
      xi = reverse ? -XC_LINE[i] : XC_LINE[i];
    }
    out[i] = -1;
    for (int n = 0; n < nloc_of(geom); n++) {
      if (xc(geom, n, d0) != sgn) continue;
      if (xc(geom, n, a) != xi) continue;
      if (d == 3 && xc(geom, n, b) != eta) continue;
      out[i] = n;
    }
  }
  return nfn;
}

}  // namespace fhfe

// ---- C-ABI -------------------------------------------------------------------------------------------------
#include "fh_internal.h"

extern "C" int fh_fe_gauss(int geom, int order, int* ng, double* w, double* x) {
  FH_REQUIRE(geom >= 0 && geom <= 4, "fh_fe_gauss: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle) or 4 (tetrahedron)");
  FH_REQUIRE(order >= 0 && order <= 4, "fh_fe_gauss: Gauss rule index %d not supported (0..4)", order);
  if (ng) *ng = fhfe::gauss_npoints(geom, order);
  if (w || x) fhfe::gauss_table(geom, order, w, x);
  return 0;
}

extern "C" int fh_fe_tables(int geom, int fe, int order, int* ng, int* nc, double* phi, double* dphi) {
  FH_REQUIRE(geom >= 0 && geom <= 4, "fh_fe_tables: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle) or 4 (tetrahedron)");
  FH_REQUIRE(geom != 4 || fe <= 1, "fh_fe_tables: on the tetrahedron the families 0 (P1) and 1 (P2, TET10) are served, not the P2 + bubble one (TET15)");
  FH_REQUIRE(fhfe::fe_known(fe), "fh_fe_tables: fe must be 0 (linear), 1 (serendipity), 2 (biquadratic) or 3 (piecewise constant)");
  FH_REQUIRE(order >= 0 && order <= 4, "fh_fe_tables: Gauss rule index %d not supported (0..4)", order);
  const int d = fhfe::dim_of(geom), n = fhfe::ndofs_of(geom, fe), g = fhfe::gauss_npoints(geom, order);
  if (ng) *ng = g;
  if (nc) *nc = n;
  if (phi || dphi) {
    std::vector<double> w, p, dp;
    fhfe::shape_tables(geom, fe, order, w, p, dp);
    if (phi) fh_copy_out(phi, p);
    if (dphi)   // reference layout: one [ng][nc] table per direction (_dphidxi, _dphideta, _dphidzeta)
      for (int k = 0; k < d; k++)
        for (int ig = 0; ig < g; ig++)
          for (int j = 0; j < n; j++) dphi[((size_t)k * g + ig) * n + j] = dp[((size_t)ig * n + j) * d + k];
  }
  return 0;
}

extern "C" int fh_fe_tables_d2(int geom, int fe, int order, double* d2phi) {
  FH_REQUIRE(geom >= 0 && geom <= 4, "fh_fe_tables_d2: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle) or 4 (tetrahedron)");
  FH_REQUIRE(geom != 4 || fe <= 1, "fh_fe_tables_d2: on the tetrahedron the families 0 (P1) and 1 (P2, TET10) are served, not the P2 + bubble one (TET15)");
  FH_REQUIRE(fhfe::fe_known(fe), "fh_fe_tables_d2: fe must be 0 (linear), 1 (serendipity), 2 (biquadratic) or 3 (piecewise constant)");
  FH_REQUIRE(order >= 0 && order <= 4 && d2phi, "fh_fe_tables_d2: bad arguments");
  const int d = fhfe::dim_of(geom), n = fhfe::ndofs_of(geom, fe), g = fhfe::gauss_npoints(geom, order), nh = d == 1 ? 1 : d == 2 ? 3 : 6;
  std::vector<double> w(g), x((size_t)g * d), t((size_t)n * nh);
  fhfe::gauss_table(geom, order, w.data(), x.data());
  for (int ig = 0; ig < g; ig++) {
    double pt[3] = {0, 0, 0};
    for (int k = 0; k < d; k++) pt[k] = x[(size_t)k * g + ig];
    fhfe::eval_basis_d2(geom, fe, pt, t.data());
    for (int k = 0; k < nh; k++)      // reference layout: one [ng][nc] table per second derivative
      for (int j = 0; j < n; j++) d2phi[((size_t)k * g + ig) * n + j] = t[(size_t)j * nh + k];
  }
  return 0;
}

extern "C" int fh_fe_elem_prolongator(int geom, int fe, int* nchild, int* nc, double* P) {
  FH_REQUIRE(geom >= 0 && geom <= 4, "fh_fe_elem_prolongator: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle) or 4 (tetrahedron)");
  FH_REQUIRE(geom != 4 || fe <= 1, "fh_fe_elem_prolongator: on the tetrahedron the families 0 (P1) and 1 (P2, TET10) are served, not the P2 + bubble one (TET15)");
  FH_REQUIRE(fhfe::fe_known(fe), "fh_fe_elem_prolongator: fe must be 0 (linear), 1 (serendipity), 2 (biquadratic) or 3 (piecewise constant)");
  if (nchild) *nchild = geom == fhfe::GEOM_TRI ? 4 : geom == fhfe::GEOM_TET ? 8 : fhfe::nvert_of(geom);
  if (nc) *nc = fhfe::ndofs_of(geom, fe);
  if (P) {
    std::vector<double> v;
    fhfe::elem_prolongator(geom, fe, v);
    fh_copy_out(P, v);
  }
  return 0;
}

// reference coordinates (-1, 0, 1 per direction) of local node `node` of the biquadratic element (hex_lag / quad_lag X tables, Hexahedron.cpp:32-92)
extern "C" int fh_fe_node_ref(int geom, int node, int* xi) {
  FH_REQUIRE(geom >= 0 && geom <= 2, "fh_fe_node_ref: geom must be 0 (hex), 1 (quad) or 2 (line) (integer coordinates; the triangle: fh_fe_node_ref_coords)");
  FH_REQUIRE(node >= 0 && node < fhfe::nloc_of(geom) && xi, "fh_fe_node_ref: node %d out of range", node);
  for (int d = 0; d < fhfe::dim_of(geom); d++) xi[d] = fhfe::xc(geom, node, d);
  return 0;
}

// reference coordinates of a local node as doubles (any element: the triangle's are 0, 1/2, 1, 1/3)
extern "C" int fh_fe_node_ref_coords(int geom, int node, double* xi) {
  FH_REQUIRE(geom >= 0 && geom <= 4 && xi, "fh_fe_node_ref_coords: bad arguments");
  FH_REQUIRE(node >= 0 && node < fhfe::nloc_of(geom), "fh_fe_node_ref_coords: node %d out of range", node);
  fhfe::node_ref(geom, node, xi);
  return 0;
}

extern "C" int fh_fe_face_nodes(int geom, int fe, int face, int* nfn, int* local_nodes) {
  FH_REQUIRE(geom >= 0 && geom <= 4, "fh_fe_face_nodes: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle) or 4 (tetrahedron)");
  FH_REQUIRE(geom != 4 || fe <= 1, "fh_fe_face_nodes: on the tetrahedron the families 0 (P1) and 1 (P2, TET10) are served, not the P2 + bubble one (TET15)");
  FH_REQUIRE(fhfe::fe_known(fe), "fh_fe_face_nodes: fe must be 0 .. 3");
  FH_REQUIRE(face >= 0 && face < fhfe::nfaces_of(geom), "fh_fe_face_nodes: face %d out of range", face);
  int tmp[9];
  const int n = fhfe::face_nodes(geom, fe, face, tmp);
  if (nfn) *nfn = n;
  if (local_nodes) memcpy(local_nodes, tmp, n * sizeof(int));
  return 0;
}
