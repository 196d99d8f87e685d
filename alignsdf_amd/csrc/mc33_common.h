// Marching Cubes 33 (Lewiner et al., JGT 8(2) 2003) case selection of the HIP kernels (mc33.hip).  Device code only: the
// sequential CPU restatement the tests check it against (oracle/mc33_oracle.c) shares nothing with this header - it has its
// own deciders and its own copy of the published tables.
//
// Behaviour being reproduced: skimage 0.18.3 `marching_cubes_lewiner` - the third-party routine the
// reference calls at utils/mesh.py:354 and deep_sdf/mesh.py:81 (its Cython source is not shipped with
// the wheel; the algorithm is restated from the paper and pinned against the installed binary by the
// goldens under tests/golden/mc_*.npz).
//
// Cube conventions (x = fastest array axis):
//   corners  v0=(x,y,z) v1=(x+1,y,z) v2=(x+1,y+1,z) v3=(x,y+1,z), v4..v7 the same at z+1
//   edges    0:v0v1 1:v1v2 2:v2v3 3:v3v0  4:v4v5 5:v5v6 6:v6v7 7:v7v4  8:v0v4 9:v1v5 10:v2v6 11:v3v7
//   vertex 12 is the interior ("centre") vertex some MC33 tilings need.
// All decisions are taken in double precision on (value - level), as the routine does.
#pragma once
#include <stdint.h>

#define MC33_HD __device__ __forceinline__

#include "mc33_tables.h"

#define MC33_EPS 2.220446049250313e-16   /* the routine's "tiny number": numpy.spacing(1.0) */

// Does face `face` (1..6, sign = orientation of the test) contain part of the surface?
// Asymptotic decider on the bilinear face interpolant.
MC33_HD int mc33_test_face(const double* v, int face) {
  const int af = face < 0 ? -face : face;
  double A, B, C, D;
  switch (af) {
    case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
    case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
    case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
    case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
    case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
    default: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;   /* 6 */
  }
  const double acbd = A * C - B * D;
  if (acbd > -MC33_EPS && acbd < MC33_EPS) return face >= 0;
  return face * A * acbd >= 0;
}

// Is the interior of the cube crossed by the surface (tunnel) for ambiguous case `cas`?
// `edge` is the reference edge of the triangulation (cases 6, 7, 12, 13), s the signed test id.
MC33_HD int mc33_test_internal(const double* v, int cas, int edge, int s) {
  double t, At = 0.0, Bt = 0.0, Ct = 0.0, Dt = 0.0, a, b;
  if (cas == 4 || cas == 10) {
    a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
    b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
    t = -b / (2 * a + MC33_EPS);
    if (t < 0 || t > 1) return s > 0;
    At = v[0] + (v[4] - v[0]) * t;
    Bt = v[3] + (v[7] - v[3]) * t;
    Ct = v[2] + (v[6] - v[2]) * t;
    Dt = v[1] + (v[5] - v[1]) * t;
  } else {
    // walk along `edge` to its iso crossing and evaluate the opposite-face interpolants there
    static const int8_t E[12][8] = {
        /* p  q   B0 B1  C0 C1  D0 D1 :  t = v[p] / (v[p] - v[q]);  Bt = v[B0] + (v[B1]-v[B0]) t ... */
        {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
        {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
        {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
    if (edge < 0 || edge > 11) return s < 0;
    const int8_t* e = E[edge];
    t = v[e[0]] / (v[e[0]] - v[e[1]] + MC33_EPS);
    At = 0.0;
    Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
    Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
    Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
  }
  int test = 0;
  if (At >= 0) test += 1;
  if (Bt >= 0) test += 2;
  if (Ct >= 0) test += 4;
  if (Dt >= 0) test += 8;
  switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    /* Pinned against the skimage 0.18.3 binary (tests/golden/mc_cells.npz): when the determinant test of
     * 5 / 10 fails the routine falls off its if-chain and yields 0 whatever the sign of s - unlike the
     * paper's code, which returns s < 0 there. */
    case 5: return (At * Ct - Bt * Dt < MC33_EPS) ? (s > 0) : 0;
    case 10: return (At * Ct - Bt * Dt >= MC33_EPS) ? (s > 0) : 0;
    default: break;   /* 7, 11, 13, 14, 15 */
  }
  return s < 0;
}

// Select the tiling of one cell.  v[0..7] = corner values minus the iso level (double).
// Returns the number of triangles (0..12) and sets *offset to the first entry of the tiling in
// kMcTiles (3 edge ids per triangle).
MC33_HD int mc33_select_tiling(const double* v, int* offset) {
  int index = 0;
  for (int c = 0; c < 8; ++c)
    if (v[c] > 0.0) index |= 1 << c;
  const int cas = kMcCases[index][0];
  const int cfg = kMcCases[index][1];
  int sub = 0;
#define MC33_PICK(table, row, nt) do { *offset = kMcOff_##table + (row) * kMcRow_##table; return (nt); } while (0)
#define MC33_PICK2(table, row, k, nt) do { *offset = kMcOff_##table + (row) * kMcRow_##table + (k) * 3 * (nt); return (nt); } while (0)
  switch (cas) {
    case 1: MC33_PICK(1, cfg, 1);
    case 2: MC33_PICK(2, cfg, 2);
    case 3:
      if (mc33_test_face(v, kMcTest3[cfg])) MC33_PICK(3_2, cfg, 4);
      MC33_PICK(3_1, cfg, 2);
    case 4:
      if (mc33_test_internal(v, cas, -1, kMcTest4[cfg])) MC33_PICK(4_1, cfg, 2);
      MC33_PICK(4_2, cfg, 6);
    case 5: MC33_PICK(5, cfg, 3);
    case 6:
      if (mc33_test_face(v, kMcTest6[cfg][0])) MC33_PICK(6_2, cfg, 5);
      if (mc33_test_internal(v, cas, kMcTest6[cfg][2], kMcTest6[cfg][1])) MC33_PICK(6_1_1, cfg, 3);
      MC33_PICK(6_1_2, cfg, 9);
    case 7:
      if (mc33_test_face(v, kMcTest7[cfg][0])) sub += 1;
      if (mc33_test_face(v, kMcTest7[cfg][1])) sub += 2;
      if (mc33_test_face(v, kMcTest7[cfg][2])) sub += 4;
      switch (sub) {
        case 0: MC33_PICK(7_1, cfg, 3);
        case 1: MC33_PICK2(7_2, cfg, 0, 5);
        case 2: MC33_PICK2(7_2, cfg, 1, 5);
        case 3: MC33_PICK2(7_3, cfg, 0, 9);
        case 4: MC33_PICK2(7_2, cfg, 2, 5);
        case 5: MC33_PICK2(7_3, cfg, 1, 9);
        case 6: MC33_PICK2(7_3, cfg, 2, 9);
        default:
          if (mc33_test_internal(v, cas, kMcTest7[cfg][4], kMcTest7[cfg][3])) MC33_PICK(7_4_2, cfg, 9);
          MC33_PICK(7_4_1, cfg, 5);
      }
    case 8: MC33_PICK(8, cfg, 2);
    case 9: MC33_PICK(9, cfg, 4);
    case 10:
      if (mc33_test_face(v, kMcTest10[cfg][0])) {
        if (mc33_test_face(v, kMcTest10[cfg][1])) MC33_PICK(10_1_1_, cfg, 4);
        MC33_PICK(10_2, cfg, 8);
      }
      if (mc33_test_face(v, kMcTest10[cfg][1])) MC33_PICK(10_2_, cfg, 8);
      if (mc33_test_internal(v, cas, -1, kMcTest10[cfg][2])) MC33_PICK(10_1_1, cfg, 4);
      MC33_PICK(10_1_2, cfg, 8);
    case 11: MC33_PICK(11, cfg, 4);
    case 12:
      if (mc33_test_face(v, kMcTest12[cfg][0])) {
        if (mc33_test_face(v, kMcTest12[cfg][1])) MC33_PICK(12_1_1_, cfg, 4);
        MC33_PICK(12_2, cfg, 8);
      }
      if (mc33_test_face(v, kMcTest12[cfg][1])) MC33_PICK(12_2_, cfg, 8);
      if (mc33_test_internal(v, cas, kMcTest12[cfg][3], kMcTest12[cfg][2])) MC33_PICK(12_1_1, cfg, 4);
      MC33_PICK(12_1_2, cfg, 8);
    case 13: {
      for (int f = 0; f < 6; ++f)
        if (mc33_test_face(v, kMcTest13[cfg][f])) sub |= 1 << f;
      sub = kMcSubconfig13[sub];
      if (sub < 0) { *offset = 0; return 0; }   /* the routine prints "impossible case 13" and emits nothing */
      if (sub == 0) MC33_PICK(13_1, cfg, 4);
      if (sub <= 6) MC33_PICK2(13_2, cfg, sub - 1, 6);
      if (sub <= 18) MC33_PICK2(13_3, cfg, sub - 7, 10);
      if (sub <= 22) MC33_PICK2(13_4, cfg, sub - 19, 12);
      if (sub <= 26) {
        const int k = sub - 23;
        const int edge = kMcTiles[kMcOff_13_5_1 + cfg * kMcRow_13_5_1 + k * 18];
        if (mc33_test_internal(v, cas, edge, kMcTest13[cfg][6])) MC33_PICK2(13_5_1, cfg, k, 6);
        MC33_PICK2(13_5_2, cfg, k, 10);
      }
      if (sub <= 38) MC33_PICK2(13_3_, cfg, sub - 27, 10);
      if (sub <= 44) MC33_PICK2(13_2_, cfg, sub - 39, 6);
      if (sub == 45) MC33_PICK(13_1_, cfg, 4);
      *offset = 0;
      return 0;   /* "impossible case 13" */
    }
    case 14: MC33_PICK(14, cfg, 4);
    default: break;
  }
#undef MC33_PICK
#undef MC33_PICK2
  *offset = 0;
  return 0;
}

// Which grid edge carries cell-local edge e, as (dx, dy, dz, axis): the edge starts at cell corner
// (x+dx, y+dy, z+dz) and runs along `axis` (0 = x, 1 = y, 2 = z).
#define MC33_EDGE_DX(e) ((0x0622 >> (e)) & 1)    /* edges 1,5,9,10 start at x+1 */
#define MC33_EDGE_DY(e) ((0x0C44 >> (e)) & 1)    /* edges 2,6,10,11 start at y+1 */
#define MC33_EDGE_DZ(e) ((0x00F0 >> (e)) & 1)    /* edges 4..7 lie in the z+1 plane */
#define MC33_EDGE_AXIS(e) ((e) >= 8 ? 2 : ((e) & 1))   /* 0,2,4,6 along x; 1,3,5,7 along y; 8..11 along z */
