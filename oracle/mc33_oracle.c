/* Sequential CPU restatement of skimage 0.18.3 `marching_cubes_lewiner`.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference calls it at utils/mesh.py:354 / deep_sdf/mesh.py:81 with level = 0.0, step_size 1,
 * allow_degenerate True, use_classic False, gradient_direction 'descent', no mask.  scikit-image is an
 * un-vendored dependency of the reference (requirements.txt:5, un-pinned; the function exists only in
 * <= 0.18.x) and ships no Cython source, so this file restates the published algorithm (Lewiner et
 * al., JGT 8(2) 2003) the way that routine organises it:
 *   - cells are visited with axis 0 slowest and axis 2 (= x) fastest;
 *   - a vertex is created the first time a cell's triangle list references a grid edge (or the
 *     cell's interior vertex) and is shared through two per-layer lookup arrays, so vertex ids follow
 *     first-reference order;
 *   - an edge vertex is the 1/(eps+|v|)-weighted mean of its two end points, evaluated in double and
 *     stored as float; the interior vertex is the same weighted mean over the 8 corners;
 *   - vertices come back in (axis0, axis1, axis2) order and every face is reversed ('descent').
 * Pinned by tests/test_oracle_mc.py against goldens produced by the installed skimage binary
 * (tests/golden/make_mc_goldens.py, run with /opt/conda/bin/python3.9).
 *
 * Build: gcc -O2 -shared -fPIC oracle/mc33_oracle.c -o oracle/_build/libmc33_oracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../alignsdf_amd/csrc/mc33_common.h"

typedef struct {
  float* verts; int nverts, cap_verts;
  int* faces; int nidx, cap_idx;
} mesh_t;

static int push_vertex(mesh_t* m, double x, double y, double z) {
  if (m->nverts == m->cap_verts) {
    m->cap_verts = m->cap_verts ? m->cap_verts * 2 : 1024;
    m->verts = (float*)realloc(m->verts, sizeof(float) * 3 * (size_t)m->cap_verts);
  }
  float* p = m->verts + 3 * (size_t)m->nverts;
  p[0] = (float)x; p[1] = (float)y; p[2] = (float)z;
  return m->nverts++;
}

static void push_index(mesh_t* m, int id) {
  if (m->nidx == m->cap_idx) {
    m->cap_idx = m->cap_idx ? m->cap_idx * 2 : 4096;
    m->faces = (int*)realloc(m->faces, sizeof(int) * (size_t)m->cap_idx);
  }
  m->faces[m->nidx++] = id;
}

/* Returns 0, -6 (level outside data range), -7 (no surface), -2 (out of memory).
 * verts_out: [V][3] floats in (axis0, axis1, axis2) voxel units; faces_out: [F][3] ints. Caller frees
 * with mc33_free. */
int mc33_lewiner(const float* vol, int n0, int n1, int n2, double level, float** verts_out, int* nverts,
                 int** faces_out, int* nfaces) {
  const int nx = n2, ny = n1, nz = n0;
  *verts_out = 0; *faces_out = 0; *nverts = 0; *nfaces = 0;
  if (nx < 2 || ny < 2 || nz < 2) return -1;
  {
    float lo = vol[0], hi = vol[0];
    const size_t n = (size_t)nx * ny * nz;
    for (size_t i = 1; i < n; ++i) { if (vol[i] < lo) lo = vol[i]; if (vol[i] > hi) hi = vol[i]; }
    if (level < lo || level > hi) return -6;
  }
  mesh_t m; memset(&m, 0, sizeof(m));
  /* 4 slots per cell position: x-edge, y-edge, z-edge, interior vertex */
  const size_t layer = (size_t)nx * ny * 4;
  int* layer1 = (int*)malloc(sizeof(int) * layer);
  int* layer2 = (int*)malloc(sizeof(int) * layer);
  if (!layer1 || !layer2) { free(layer1); free(layer2); return -2; }
  for (size_t i = 0; i < layer; ++i) layer1[i] = layer2[i] = -1;

  for (int z = 0; z < nz - 1; ++z) {
    { int* t = layer1; layer1 = layer2; layer2 = t; }
    for (size_t i = 0; i < layer; ++i) layer2[i] = -1;
    for (int y = 0; y < ny - 1; ++y) {
      for (int x = 0; x < nx - 1; ++x) {
        const float* p0 = vol + ((size_t)z * ny + y) * nx + x;
        const float* p1 = p0 + (size_t)ny * nx;
        double v[8];
        v[0] = (double)p0[0] - level; v[1] = (double)p0[1] - level;
        v[2] = (double)p0[nx + 1] - level; v[3] = (double)p0[nx] - level;
        v[4] = (double)p1[0] - level; v[5] = (double)p1[1] - level;
        v[6] = (double)p1[nx + 1] - level; v[7] = (double)p1[nx] - level;
        int off;
        const int nt = mc33_select_tiling(v, &off);
        if (nt == 0) continue;
        double cx = 0, cy = 0, cz = 0; int have_centre = 0;
        for (int k = 0; k < 3 * nt; ++k) {
          const int e = kMcTiles[off + k];
          int* slot;
          if (e == 12) {
            slot = &layer1[4 * ((size_t)nx * y + x) + 3];
          } else {
            const int ex = x + MC33_EDGE_DX(e), ey = y + MC33_EDGE_DY(e);
            int* lay = MC33_EDGE_DZ(e) ? layer2 : layer1;
            slot = &lay[4 * ((size_t)nx * ey + ex) + MC33_EDGE_AXIS(e)];
          }
          if (*slot < 0) {
            if (e == 12) {
              if (!have_centre) {
                /* corner offsets in v0..v7 order */
                static const double ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1},
                                    oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
                double fx = 0, fy = 0, fz = 0, ff = 0;
                for (int c = 0; c < 8; ++c) {
                  const double w = 1.0 / (MC33_EPS + fabs(v[c]));
                  fx += ox[c] * w; fy += oy[c] * w; fz += oz[c] * w; ff += w;
                }
                cx = x + fx / ff; cy = y + fy / ff; cz = z + fz / ff;
                have_centre = 1;
              }
              *slot = push_vertex(&m, cx, cy, cz);
            } else {
              /* end points of edge e as corner ids */
              static const int8_t ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
              static const int8_t eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
              static const double ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1},
                                  oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
              const int a = ea[e], b = eb[e];
              const double wa = 1.0 / (MC33_EPS + fabs(v[a])), wb = 1.0 / (MC33_EPS + fabs(v[b]));
              double fx = 0, fy = 0, fz = 0, ff = 0;
              fx += ox[a] * wa; fy += oy[a] * wa; fz += oz[a] * wa; ff += wa;
              fx += ox[b] * wb; fy += oy[b] * wb; fz += oz[b] * wb; ff += wb;
              *slot = push_vertex(&m, x + fx / ff, y + fy / ff, z + fz / ff);
            }
          }
          push_index(&m, *slot);
        }
      }
    }
  }
  free(layer1); free(layer2);
  if (m.nverts == 0) { free(m.verts); free(m.faces); return -7; }
  /* (x,y,z) -> (axis0, axis1, axis2) and reverse every face */
  for (int i = 0; i < m.nverts; ++i) { float* p = m.verts + 3 * (size_t)i; float t = p[0]; p[0] = p[2]; p[2] = t; }
  for (int f = 0; f < m.nidx / 3; ++f) { int* q = m.faces + 3 * (size_t)f; int t = q[0]; q[0] = q[2]; q[2] = t; }
  *verts_out = m.verts; *nverts = m.nverts; *faces_out = m.faces; *nfaces = m.nidx / 3;
  return 0;
}

void mc33_free(void* p) { free(p); }
