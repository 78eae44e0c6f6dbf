/*
 * icp.c -- the build-defined scan-to-map registration.
 * TEST INFRASTRUCTURE ONLY (see velo_oracle.h).
 *
 * PARITY UNPINNED w.r.t. the reference: victl/VeloSLAM contains no ICP, no kNN,
 * no voxel grid, no normal estimation and no linear solve (the only
 * trace is an unused #include <pcl/kdtree/kdtree_flann.h>, HDLParser.h:66).
 * This file IS the specification; DESIGN.md section "ICP semantics" states it in
 * prose.  The HIP kernels are held to it:
 *   - map sort order, cell table, normals, correspondences: BIT-EXACT
 *     (integer/index work, and float arithmetic written with explicit
 *     fmaf()/fma() under -ffp-contract=off on both sides);
 *   - JtJ / Jtr sums and the pose: <= 1e-4 m, 1e-5 rad (summation order differs).
 *
 * Semantics
 *  grid     origin o = component-wise min of the map points (float) unless an explicit grid
 *           is given (vo_map_build_grid; rolling map rules further down); inv_h =
 *           1.0f/h; u = (p - o) * inv_h; voxel c = floorf(u) per axis; dims = c(max)+1.
 *           Each voxel is split into S x S x S sub-cells (S = 3 by default):
 *           s = min(S-1, floorf((u - c) * S)), fine coordinate F = c*S + s, fine
 *           key = (Fz*NFy + Fy)*NFx + Fx with NF = S*dims.  Points are STABLY sorted
 *           by fine key; fine_start[k] = number of keys < k.  "Sorted index" is the
 *           position in this order.  A row of fine cells (fixed Fy,Fz) is one
 *           contiguous index range; a voxel is S*S such row pieces.
 *           The CANDIDATE SET of a query is still defined on voxels: all points of
 *           the 27 voxels around the query's voxel (81 fine rows for S = 3).
 *  normals  for sorted point s: the k smallest (d2, append-order index) among all points
 *           of the 27 neighbouring cells with d2 <= (0.99 h)^2 (self included) -- that ball
 *           lies strictly inside the 27 voxels wherever the grid is anchored, float rounding
 *           of the cell assignment included, so a normal is a function of the point list alone; fewer than 5 -> normal = 0 (invalid).  Covariance
 *           about the mean in fp64, summed in that (d2, append index) order; eigenvector of the smallest
 *           eigenvalue by 8 fixed cyclic Jacobi sweeps (only + - * / sqrt);
 *           sign: last non-zero of (nz, ny, nx) made positive; stored as float.
 *  kNN      q = (float)(T*p) with T*p in fp64 by nested fma; exhaustive scan of
 *           the 27 cells in ascending sorted index; d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx))
 *           with d = candidate - q; best = first minimum (ties -> lowest sorted
 *           index); valid iff d2 <= d_max*d_max (float) and d_max <= h.
 *  residual r = n.(p' - q_map), J = [p' x n, n] (rotation first, left
 *           perturbation), p' in fp64; pairs whose map normal is invalid are
 *           dropped; 21 + 6 + 1 + 1 = 29 doubles.
 *  solve    LDLt without pivoting; a non-positive pivot -> retry once with
 *           H + 1e-9 I; fewer than 6 pairs -> no update.  T <- exp(xi^) T.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "velo_oracle.h"

struct vo_map {
    size_t n, ncell; /* ncell = number of FINE cells */
    float o[3], inv_h, h;
    int dims[3];  /* voxels (coarse cells) per axis */
    int S;        /* sub-cells per voxel edge */
    int fd[3];    /* fine cells per axis = S * dims */
    float *x, *y, *z, *nx, *ny, *nz;
    int32_t* perm;
    int32_t* cell_start;
};

static inline int cell_coord(float p, float o, float inv_h, int dim)
{
    float f = floorf((p - o) * inv_h);
    /* clamp before the float->int conversion; anything outside [-2, dim+1]
     * has no neighbour inside the grid anyway */
    if (!(f >= -2.0f)) f = -2.0f;
    if (f > (float)(dim + 1)) f = (float)(dim + 1);
    return (int)f;
}

/* fine coordinate of a MAP point (always inside the grid) */
static inline int fine_coord(float p, float o, float inv_h, int S)
{
    const float u = (p - o) * inv_h;
    const float c = floorf(u);
    int sub = (int)floorf((u - c) * (float)S);
    if (sub > S - 1) sub = S - 1;
    if (sub < 0) sub = 0;
    return (int)c * S + sub;
}

/* Iterate the fine rows of the 3x3x3 voxel block around voxel (cx,cy,cz) in ascending
 * sorted index.  Usage: ROWS_BEGIN(m,cx,cy,cz) { ... j0, j1 ... } ROWS_END */
#define ROWS_BEGIN(m, cx, cy, cz)                                                              \
    {                                                                                          \
        const int vx0_ = (cx)-1 < 0 ? 0 : (cx)-1,                                              \
                  vx1_ = (cx) + 1 >= (m)->dims[0] ? (m)->dims[0] - 1 : (cx) + 1;              \
        const int vy0_ = (cy)-1 < 0 ? 0 : (cy)-1,                                              \
                  vy1_ = (cy) + 1 >= (m)->dims[1] ? (m)->dims[1] - 1 : (cy) + 1;              \
        const int vz0_ = (cz)-1 < 0 ? 0 : (cz)-1,                                              \
                  vz1_ = (cz) + 1 >= (m)->dims[2] ? (m)->dims[2] - 1 : (cz) + 1;              \
        if (vx0_ <= vx1_ && vy0_ <= vy1_ && vz0_ <= vz1_)                                      \
            for (int fz_ = vz0_ * (m)->S; fz_ < (vz1_ + 1) * (m)->S; ++fz_)                   \
                for (int fy_ = vy0_ * (m)->S; fy_ < (vy1_ + 1) * (m)->S; ++fy_) {             \
                    const size_t row_ = ((size_t)fz_ * (m)->fd[1] + fy_) * (m)->fd[0];        \
                    const int32_t j0 = (m)->cell_start[row_ + (size_t)vx0_ * (m)->S];         \
                    const int32_t j1 = (m)->cell_start[row_ + (size_t)(vx1_ + 1) * (m)->S];
#define ROWS_END \
    }            \
    }

/* ---- symmetric 3x3 eigen decomposition: fixed cyclic Jacobi, fp64 ---------- */
static void jacobi_rot(double A[3][3], double V[3][3], int p, int q)
{
    if (A[p][q] == 0.0) return;
    double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
    if (theta < 0.0) t = -t;
    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    int r = 3 - p - q;
    double app = A[p][p], aqq = A[q][q], apq = A[p][q];
    double arp = A[r][p], arq = A[r][q];
    A[p][p] = app - t * apq;
    A[q][q] = aqq + t * apq;
    A[p][q] = A[q][p] = 0.0;
    A[r][p] = A[p][r] = c * arp - s * arq;
    A[r][q] = A[q][r] = s * arp + c * arq;
    for (int k = 0; k < 3; ++k) {
        double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}

static void smallest_eigvec(const double C[6] /* xx xy xz yy yz zz */, double n[3])
{
    double A[3][3] = {{C[0], C[1], C[2]}, {C[1], C[3], C[4]}, {C[2], C[4], C[5]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 8; ++sweep) {
        jacobi_rot(A, V, 0, 1);
        jacobi_rot(A, V, 0, 2);
        jacobi_rot(A, V, 1, 2);
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    double vx = V[0][m], vy = V[1][m], vz = V[2][m];
    double inv = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
    vx *= inv;
    vy *= inv;
    vz *= inv;
    int flip = (vz < 0.0) || (vz == 0.0 && (vy < 0.0 || (vy == 0.0 && vx < 0.0)));
    if (flip) {
        vx = -vx;
        vy = -vy;
        vz = -vz;
    }
    n[0] = vx;
    n[1] = vy;
    n[2] = vz;
}

#define VO_KMAX 32
#define VO_NORMAL_RADIUS 0.99f
#define VO_MIN_NB 5

static void point_normal(const vo_map* m, size_t s, int k, float out[3])
{
    const float qx = m->x[s], qy = m->y[s], qz = m->z[s];
    /* neighbour radius 0.99 h, strictly inside one voxel: a neighbour's voxel index then
     * differs by at most one from the point's on every axis whatever the float rounding of the
     * cell assignment does (up to ~4e4 voxels per axis), so the neighbour set -- and with the
     * append-order tie-break the whole normal -- does not depend on where the grid is anchored */
    const float rn = VO_NORMAL_RADIUS * m->h;
    const float r2 = rn * rn;
    const int cx = cell_coord(qx, m->o[0], m->inv_h, m->dims[0]);
    const int cy = cell_coord(qy, m->o[1], m->inv_h, m->dims[1]);
    const int cz = cell_coord(qz, m->o[2], m->inv_h, m->dims[2]);
    float bd[VO_KMAX];
    int32_t bi[VO_KMAX];
    int cnt = 0;
    ROWS_BEGIN(m, cx, cy, cz)
    {
        for (int32_t j = j0; j < j1; ++j) {
            float dx = m->x[j] - qx, dyy = m->y[j] - qy, dzz = m->z[j] - qz;
            float d2 = fmaf(dzz, dzz, fmaf(dyy, dyy, dx * dx));
            if (!(d2 <= r2)) continue;
            /* insert into the ascending (d2, append-order index) list of at most k.  Ties are
             * broken by the APPEND-ORDER index perm[j], not the sorted index, so that a
             * normal depends on the point list only and not on where the grid is anchored */
#define NB_BEFORE(d2_, j_, i_) \
    ((d2_) < bd[i_] || ((d2_) == bd[i_] && m->perm[j_] < m->perm[bi[i_]]))
            if (cnt == k && !NB_BEFORE(d2, j, k - 1)) continue;
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && NB_BEFORE(d2, j, pos - 1)) {
                bd[pos] = bd[pos - 1];
                bi[pos] = bi[pos - 1];
                --pos;
            }
#undef NB_BEFORE
            bd[pos] = d2;
            bi[pos] = j;
            if (cnt < k) ++cnt;
        }
    }
    ROWS_END
    if (cnt < VO_MIN_NB) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    double mx = 0, my = 0, mz = 0;
    for (int i = 0; i < cnt; ++i) {
        mx += (double)m->x[bi[i]];
        my += (double)m->y[bi[i]];
        mz += (double)m->z[bi[i]];
    }
    const double invn = 1.0 / (double)cnt;
    mx *= invn;
    my *= invn;
    mz *= invn;
    double C[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < cnt; ++i) {
        double dx = (double)m->x[bi[i]] - mx, dy = (double)m->y[bi[i]] - my,
               dz = (double)m->z[bi[i]] - mz;
        C[0] = fma(dx, dx, C[0]);
        C[1] = fma(dx, dy, C[1]);
        C[2] = fma(dx, dz, C[2]);
        C[3] = fma(dy, dy, C[3]);
        C[4] = fma(dy, dz, C[4]);
        C[5] = fma(dz, dz, C[5]);
    }
    double nrm[3];
    smallest_eigvec(C, nrm);
    out[0] = (float)nrm[0];
    out[1] = (float)nrm[1];
    out[2] = (float)nrm[2];
}

vo_map* vo_map_build(const float* x, const float* y, const float* z, size_t n, float voxel,
                     int k_normals)
{
    return vo_map_build_ex(x, y, z, n, voxel, k_normals, 3);
}

/* Sub-division chosen from the map's density (subdiv == 0 in the calls below): rho = points per
 * OCCUPIED voxel (voxels anchored on the component-wise minimum; the count does not depend on a
 * whole-voxel margin), S = round(max(1.6 rho^0.2, 1.137 rho^0.314)) clamped to [2, 8] -- a fit of
 * the measured optima: S = 3 at rho = 22 (1 M-point map), 4 at 67 (3 M), 6 at 200-220 (9-10 M; 5
 * until round 3, when bounds taken before the block search moved the optimum of dense maps up:
 * DESIGN.md): finer cells mean fewer candidates per cell but more stragglers while the pose is
 * still off, and a larger table.  Resolved once, when a map is reset. */
int vo_auto_subdiv(const float* x, const float* y, const float* z, size_t n, float voxel)
{
    if (n == 0 || !(voxel > 0)) return 3;
    const float inv_h = 1.0f / voxel;
    float mn[3] = {x[0], y[0], z[0]}, mx[3] = {x[0], y[0], z[0]};
    for (size_t i = 1; i < n; ++i) {
        if (x[i] < mn[0]) mn[0] = x[i];
        if (y[i] < mn[1]) mn[1] = y[i];
        if (z[i] < mn[2]) mn[2] = z[i];
        if (x[i] > mx[0]) mx[0] = x[i];
        if (y[i] > mx[1]) mx[1] = y[i];
        if (z[i] > mx[2]) mx[2] = z[i];
    }
    size_t d[3], nv = 1;
    for (int a = 0; a < 3; ++a) {
        d[a] = (size_t)floorf((mx[a] - mn[a]) * inv_h) + 1;
        nv *= d[a];
    }
    if (nv >= ((size_t)1 << 31)) return 1;
    unsigned char* occ = (unsigned char*)calloc(nv, 1);
    size_t n_occ = 0;
    for (size_t i = 0; i < n; ++i) {
        const size_t cx = (size_t)floorf((x[i] - mn[0]) * inv_h), cy = (size_t)floorf((y[i] - mn[1]) * inv_h),
                     cz = (size_t)floorf((z[i] - mn[2]) * inv_h);
        unsigned char* o = &occ[(cz * d[1] + cy) * d[0] + cx];
        n_occ += (size_t)(*o == 0);
        *o = 1;
    }
    free(occ);
    const double rho = (double)n / (double)n_occ;
    const double a = 1.6 * pow(rho, 0.2), b = 1.137 * pow(rho, 0.314);
    int S = (int)floor((a > b ? a : b) + 0.5);
    return S < 2 ? 2 : (S > 8 ? 8 : S);
}

vo_map* vo_map_build_ex(const float* x, const float* y, const float* z, size_t n, float voxel,
                        int k_normals, int subdiv)
{
    if (subdiv == 0) subdiv = vo_auto_subdiv(x, y, z, n, voxel);
    return vo_map_build_grid(x, y, z, n, voxel, k_normals, subdiv, NULL, NULL);
}

/* Fresh build on an explicit grid: origin (must not exceed the min of the points on any
 * axis; NULL = the min itself) and a lower bound on the voxel dims (NULL = tight).  The
 * rolling map below is DEFINED as "fresh build of the current raw list on the current grid". */
vo_map* vo_map_build_grid(const float* x, const float* y, const float* z, size_t n, float voxel,
                          int k_normals, int subdiv, const float* origin, const int* dims_min)
{
    if (n == 0 || !(voxel > 0) || k_normals > VO_KMAX || subdiv < 1 || subdiv > 16) return NULL;
    vo_map* m = (vo_map*)calloc(1, sizeof *m);
    m->n = n;
    m->S = subdiv;
    m->h = voxel;
    m->inv_h = 1.0f / voxel;
    float mn[3] = {x[0], y[0], z[0]}, mxv[3] = {x[0], y[0], z[0]};
    for (size_t i = 1; i < n; ++i) {
        if (x[i] < mn[0]) mn[0] = x[i];
        if (y[i] < mn[1]) mn[1] = y[i];
        if (z[i] < mn[2]) mn[2] = z[i];
        if (x[i] > mxv[0]) mxv[0] = x[i];
        if (y[i] > mxv[1]) mxv[1] = y[i];
        if (z[i] > mxv[2]) mxv[2] = z[i];
    }
    for (int a = 0; a < 3; ++a) {
        if (origin && origin[a] > mn[a]) {
            free(m);
            return NULL;
        }
        m->o[a] = origin ? origin[a] : mn[a];
        m->dims[a] = (int)floorf((mxv[a] - m->o[a]) * m->inv_h) + 1;
        if (dims_min && dims_min[a] > m->dims[a]) m->dims[a] = dims_min[a];
        m->fd[a] = m->dims[a] * m->S;
    }
    m->ncell = (size_t)m->fd[0] * m->fd[1] * m->fd[2];
    if (m->ncell >= ((size_t)1 << 31)) {
        free(m);
        return NULL;
    }
    int32_t* key = (int32_t*)malloc(n * sizeof(int32_t));
    m->cell_start = (int32_t*)calloc(m->ncell + 1, sizeof(int32_t));
    for (size_t i = 0; i < n; ++i) {
        int cx = fine_coord(x[i], m->o[0], m->inv_h, m->S);
        int cy = fine_coord(y[i], m->o[1], m->inv_h, m->S);
        int cz = fine_coord(z[i], m->o[2], m->inv_h, m->S);
        key[i] = (int32_t)(((size_t)cz * m->fd[1] + cy) * m->fd[0] + cx);
        m->cell_start[key[i] + 1]++;
    }
    for (size_t c = 0; c < m->ncell; ++c) m->cell_start[c + 1] += m->cell_start[c];
    int32_t* cursor = (int32_t*)malloc(m->ncell * sizeof(int32_t));
    memcpy(cursor, m->cell_start, m->ncell * sizeof(int32_t));
    m->perm = (int32_t*)malloc(n * sizeof(int32_t));
    m->x = (float*)malloc(n * sizeof(float));
    m->y = (float*)malloc(n * sizeof(float));
    m->z = (float*)malloc(n * sizeof(float));
    m->nx = (float*)malloc(n * sizeof(float));
    m->ny = (float*)malloc(n * sizeof(float));
    m->nz = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) { /* stable counting sort */
        int32_t s = cursor[key[i]]++;
        m->perm[s] = (int32_t)i;
        m->x[s] = x[i];
        m->y[s] = y[i];
        m->z[s] = z[i];
    }
    free(cursor);
    free(key);
    if (k_normals > 0) {
#pragma omp parallel for schedule(dynamic, 1024)
        for (long s = 0; s < (long)n; ++s) {
            float nn[3];
            point_normal(m, (size_t)s, k_normals, nn);
            m->nx[s] = nn[0];
            m->ny[s] = nn[1];
            m->nz[s] = nn[2];
        }
    } else {
        memset(m->nx, 0, n * sizeof(float));
        memset(m->ny, 0, n * sizeof(float));
        memset(m->nz, 0, n * sizeof(float));
    }
    return m;
}

void vo_map_free(vo_map* m)
{
    if (!m) return;
    free(m->x);
    free(m->y);
    free(m->z);
    free(m->nx);
    free(m->ny);
    free(m->nz);
    free(m->perm);
    free(m->cell_start);
    free(m);
}
size_t vo_map_size(const vo_map* m) { return m->n; }
int vo_map_subdiv(const vo_map* m) { return m->S; }
void vo_map_grid(const vo_map* m, float origin[3], int dims[3], float* inv_h)
{
    memcpy(origin, m->o, sizeof m->o);
    memcpy(dims, m->dims, sizeof m->dims);
    *inv_h = m->inv_h;
}
const float* vo_map_x(const vo_map* m) { return m->x; }
const float* vo_map_y(const vo_map* m) { return m->y; }
const float* vo_map_z(const vo_map* m) { return m->z; }
const float* vo_map_nx(const vo_map* m) { return m->nx; }
const float* vo_map_ny(const vo_map* m) { return m->ny; }
const float* vo_map_nz(const vo_map* m) { return m->nz; }
const int32_t* vo_map_perm(const vo_map* m) { return m->perm; }
const int32_t* vo_map_cell_start(const vo_map* m) { return m->cell_start; }
size_t vo_map_num_cells(const vo_map* m) { return m->ncell; }


/* ---- rolling map (BASELINE configs[2]) ---------------------------------------
 * State: the raw point list in insertion order, a sticky grid (origin o, voxel dims) and a
 * margin of M voxels (one value per axis: a vehicle needs slack in x/y, hardly any in z).  After every operation the map equals
 * vo_map_build_grid(raw list, o, dims): the GPU's incremental update is held to that.
 *   anchor   o = min - M*h (float), dims = floorf((max - o)*inv_h) + 1 + M
 *   append   if a new point lies below o on some axis -> anchor on the whole list; else o
 *            stays and an axis whose needed dims floorf((max - o)*inv_h)+1 exceed dims grows
 *            to needed + M
 *   evict    keep the points with lo <= p <= hi on every axis (order preserving); nothing
 *            kept -> refused, map unchanged; if floorf((min - o)*inv_h) >= 2M+2 on some axis
 *            -> anchor; else the grid stays as it is */
struct vo_roll {
    float *x, *y, *z;
    size_t n, cap;
    float h;
    int k, S, M[3]; /* margin per axis, voxels */
    float o[3];
    int dims[3];
    vo_map* map;
};

static void roll_minmax(const vo_roll* r, float mn[3], float mx[3])
{
    mn[0] = mx[0] = r->x[0];
    mn[1] = mx[1] = r->y[0];
    mn[2] = mx[2] = r->z[0];
    for (size_t i = 1; i < r->n; ++i) {
        if (r->x[i] < mn[0]) mn[0] = r->x[i];
        if (r->y[i] < mn[1]) mn[1] = r->y[i];
        if (r->z[i] < mn[2]) mn[2] = r->z[i];
        if (r->x[i] > mx[0]) mx[0] = r->x[i];
        if (r->y[i] > mx[1]) mx[1] = r->y[i];
        if (r->z[i] > mx[2]) mx[2] = r->z[i];
    }
}

static void roll_anchor(vo_roll* r, const float mn[3], const float mx[3])
{
    const float inv_h = 1.0f / r->h;
    for (int a = 0; a < 3; ++a) {
        r->o[a] = mn[a] - (float)r->M[a] * r->h;
        r->dims[a] = (int)floorf((mx[a] - r->o[a]) * inv_h) + 1 + r->M[a];
    }
}

static int roll_rebuild(vo_roll* r)
{
    vo_map* m = vo_map_build_grid(r->x, r->y, r->z, r->n, r->h, r->k, r->S, r->o, r->dims);
    if (!m) return -1;
    vo_map_free(r->map);
    r->map = m;
    return 0;
}

static void roll_reserve(vo_roll* r, size_t n)
{
    if (n <= r->cap) return;
    size_t c = r->cap ? r->cap : 1024;
    while (c < n) c *= 2;
    r->x = (float*)realloc(r->x, c * sizeof(float));
    r->y = (float*)realloc(r->y, c * sizeof(float));
    r->z = (float*)realloc(r->z, c * sizeof(float));
    r->cap = c;
}

vo_roll* vo_roll_new(const float* x, const float* y, const float* z, size_t n, float voxel,
                     int k_normals, int subdiv, int margin)
{
    const int m3[3] = {margin, margin, margin};
    return vo_roll_new3(x, y, z, n, voxel, k_normals, subdiv, m3);
}

vo_roll* vo_roll_new3(const float* x, const float* y, const float* z, size_t n, float voxel,
                      int k_normals, int subdiv, const int margin[3])
{
    if (n == 0 || margin[0] < 0 || margin[1] < 0 || margin[2] < 0) return NULL;
    vo_roll* r = (vo_roll*)calloc(1, sizeof *r);
    r->h = voxel;
    r->k = k_normals;
    r->S = subdiv ? subdiv : vo_auto_subdiv(x, y, z, n, voxel);  /* resolved once, kept for life */
    memcpy(r->M, margin, sizeof r->M);
    roll_reserve(r, n);
    memcpy(r->x, x, n * sizeof(float));
    memcpy(r->y, y, n * sizeof(float));
    memcpy(r->z, z, n * sizeof(float));
    r->n = n;
    float mn[3], mx[3];
    roll_minmax(r, mn, mx);
    roll_anchor(r, mn, mx);
    if (roll_rebuild(r)) {
        vo_roll_free(r);
        return NULL;
    }
    return r;
}

void vo_roll_free(vo_roll* r)
{
    if (!r) return;
    vo_map_free(r->map);
    free(r->x);
    free(r->y);
    free(r->z);
    free(r);
}

const vo_map* vo_roll_map(const vo_roll* r) { return r->map; }
size_t vo_roll_size(const vo_roll* r) { return r->n; }

int vo_roll_append(vo_roll* r, const float* x, const float* y, const float* z, size_t m)
{
    if (m == 0) return 0;
    roll_reserve(r, r->n + m);
    memcpy(r->x + r->n, x, m * sizeof(float));
    memcpy(r->y + r->n, y, m * sizeof(float));
    memcpy(r->z + r->n, z, m * sizeof(float));
    int below = 0;
    for (size_t i = 0; i < m; ++i)
        if (x[i] < r->o[0] || y[i] < r->o[1] || z[i] < r->o[2]) below = 1;
    r->n += m;
    float mn[3], mx[3];
    roll_minmax(r, mn, mx);
    if (below) {
        roll_anchor(r, mn, mx);
    } else {
        const float inv_h = 1.0f / r->h;
        for (int a = 0; a < 3; ++a) {
            const int need = (int)floorf((mx[a] - r->o[a]) * inv_h) + 1;
            if (need > r->dims[a]) r->dims[a] = need + r->M[a];
        }
    }
    if (roll_rebuild(r)) return -1;
    return below;
}

/* Voxel-downsampled insertion: of the m new points, taken IN ORDER, one is
 * accepted iff its voxel (on the map's current grid, which extends to any integer coordinate)
 * holds fewer than min_count points counting the map's and those accepted before it.  What a
 * sequence of frames integrated one after another would leave behind; it keeps F frames x W
 * ranks that all see the same under-filled voxel from piling near-duplicates into it. */
static inline int sparse_coord(float p, float o, float inv_h)
{
    float f = floorf((p - o) * inv_h);
    f = f < -1048576.0f ? -1048576.0f : (f > 1048575.0f ? 1048575.0f : f);
    return (int)f;
}

size_t vo_roll_filter_sparse(const vo_roll* r, const float* x, const float* y, const float* z,
                             size_t m, int min_count, unsigned char* accept)
{
    const vo_map* mp = r->map;
    size_t cap = 16;
    while (cap < 2 * m + 1) cap <<= 1;
    uint64_t* keys = (uint64_t*)malloc(cap * sizeof(uint64_t));
    int32_t* cnt = (int32_t*)calloc(cap, sizeof(int32_t));
    memset(keys, 0xFF, cap * sizeof(uint64_t));
    size_t n_acc = 0;
    for (size_t i = 0; i < m; ++i) {
        const int cx = sparse_coord(x[i], mp->o[0], mp->inv_h), cy = sparse_coord(y[i], mp->o[1], mp->inv_h),
                  cz = sparse_coord(z[i], mp->o[2], mp->inv_h);
        const uint64_t key = ((uint64_t)(cz + 1048576) << 42) | ((uint64_t)(cy + 1048576) << 21) |
                             (uint64_t)(cx + 1048576);
        size_t hsh = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
        while (keys[hsh] != key && keys[hsh] != UINT64_MAX) hsh = (hsh + 1) & (cap - 1);
        if (keys[hsh] == UINT64_MAX) {  /* first new point of this voxel: start from the map's count */
            keys[hsh] = key;
            int occ = 0;
            if (cx >= 0 && cx < mp->dims[0] && cy >= 0 && cy < mp->dims[1] && cz >= 0 && cz < mp->dims[2])
                for (int fz = cz * mp->S; fz < (cz + 1) * mp->S; ++fz)
                    for (int fy = cy * mp->S; fy < (cy + 1) * mp->S; ++fy) {
                        size_t row = ((size_t)fz * mp->fd[1] + fy) * mp->fd[0];
                        occ += mp->cell_start[row + (size_t)(cx + 1) * mp->S] -
                               mp->cell_start[row + (size_t)cx * mp->S];
                    }
            cnt[hsh] = occ;
        }
        accept[i] = cnt[hsh] < min_count;
        if (accept[i]) {
            cnt[hsh]++;
            ++n_acc;
        }
    }
    free(keys);
    free(cnt);
    return n_acc;
}

/* filter, then vo_roll_append of the survivors; returns the number appended (-1 on failure) */
long vo_roll_append_sparse(vo_roll* r, const float* x, const float* y, const float* z, size_t m,
                           int min_count)
{
    if (m == 0) return 0;
    unsigned char* acc = (unsigned char*)malloc(m);
    const size_t k = vo_roll_filter_sparse(r, x, y, z, m, min_count, acc);
    float *ax = (float*)malloc((k + 1) * sizeof(float)), *ay = (float*)malloc((k + 1) * sizeof(float)),
          *az = (float*)malloc((k + 1) * sizeof(float));
    size_t w = 0;
    for (size_t i = 0; i < m; ++i)
        if (acc[i]) {
            ax[w] = x[i];
            ay[w] = y[i];
            az[w] = z[i];
            ++w;
        }
    const int rc = k ? vo_roll_append(r, ax, ay, az, k) : 0;
    free(acc);
    free(ax);
    free(ay);
    free(az);
    return rc < 0 ? -1 : (long)k;
}

/* keep region = closed box, optionally intersected with the vertical cylinder of radius `radius`
 * around (cx, cy) (radius < 0: box only).  The cylinder is the rolling-map policy behind
 * ROI_RANGE (MapManager.h:13: "sensor detecting range"): distance in the ground plane, z free. */
static inline int roll_keep(const vo_roll* r, size_t i, const float lo[3], const float hi[3],
                            float cx, float cy, float radius)
{
    if (!(r->x[i] >= lo[0] && r->x[i] <= hi[0] && r->y[i] >= lo[1] && r->y[i] <= hi[1] &&
          r->z[i] >= lo[2] && r->z[i] <= hi[2]))
        return 0;
    if (radius < 0.0f) return 1;
    const float dx = r->x[i] - cx, dy = r->y[i] - cy;
    return fmaf(dy, dy, dx * dx) <= radius * radius;
}

int vo_roll_evict_region(vo_roll* r, const float lo[3], const float hi[3], float cx, float cy,
                         float radius)
{
    size_t kept = 0;
    for (size_t i = 0; i < r->n; ++i) kept += (size_t)roll_keep(r, i, lo, hi, cx, cy, radius);
    if (kept == 0) return -1;
    if (kept == r->n) return 0;
    size_t w = 0;
    for (size_t i = 0; i < r->n; ++i)
        if (roll_keep(r, i, lo, hi, cx, cy, radius)) {
            r->x[w] = r->x[i];
            r->y[w] = r->y[i];
            r->z[w] = r->z[i];
            ++w;
        }
    r->n = w;
    float mn[3], mx[3];
    roll_minmax(r, mn, mx);
    const float inv_h = 1.0f / r->h;
    int anchor = 0;
    for (int a = 0; a < 3; ++a)
        if (floorf((mn[a] - r->o[a]) * inv_h) >= (float)(2 * r->M[a] + 2)) anchor = 1;
    if (anchor) roll_anchor(r, mn, mx);
    if (roll_rebuild(r)) return -1;
    return anchor ? 1 : 2;
}

int vo_roll_evict_outside(vo_roll* r, const float lo[3], const float hi[3])
{
    return vo_roll_evict_region(r, lo, hi, 0.0f, 0.0f, -1.0f);
}

/* p' = T*p in fp64 with a fixed fma nesting (shared with the HIP kernels) */
static inline void xform(const double T[12], float x, float y, float z, double o[3])
{
    const double dx = x, dy = y, dz = z;
    o[0] = fma(T[0], dx, fma(T[1], dy, fma(T[2], dz, T[3])));
    o[1] = fma(T[4], dx, fma(T[5], dy, fma(T[6], dz, T[7])));
    o[2] = fma(T[8], dx, fma(T[9], dy, fma(T[10], dz, T[11])));
}

static inline int32_t nearest(const vo_map* m, float qx, float qy, float qz, float* best_d2,
                              uint64_t* scanned)
{
    const int cx = cell_coord(qx, m->o[0], m->inv_h, m->dims[0]);
    const int cy = cell_coord(qy, m->o[1], m->inv_h, m->dims[1]);
    const int cz = cell_coord(qz, m->o[2], m->inv_h, m->dims[2]);
    float bd = INFINITY;
    int32_t bj = -1;
    ROWS_BEGIN(m, cx, cy, cz)
    {
        *scanned += (uint64_t)(j1 - j0);
        for (int32_t j = j0; j < j1; ++j) {
            float dx = m->x[j] - qx, dyy = m->y[j] - qy, dzz = m->z[j] - qz;
            float d2 = fmaf(dzz, dzz, fmaf(dyy, dyy, dx * dx));
            if (d2 < bd) {
                bd = d2;
                bj = j;
            }
        }
    }
    ROWS_END
    *best_d2 = bd;
    return bj;
}

uint64_t vo_correspond(const vo_map* m, const float* x, const float* y, const float* z, size_t n,
                       const double T[12], float d_max, int32_t* corr, float* d2out)
{
    const float dmax2 = d_max * d_max;
    uint64_t total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (long i = 0; i < (long)n; ++i) {
        double p[3];
        xform(T, x[i], y[i], z[i], p);
        float bd;
        uint64_t sc = 0;
        int32_t j = nearest(m, (float)p[0], (float)p[1], (float)p[2], &bd, &sc);
        total += sc;
        if (!(j >= 0 && bd <= dmax2)) {
            j = -1;
        }
        corr[i] = j;
        if (d2out) d2out[i] = (j >= 0) ? bd : INFINITY;
    }
    return total;
}

/* a10, k > 1: the k smallest (d2, sorted index) with d2 <= d_max^2 among the 27 voxels */
void vo_knn(const vo_map* m, const float* x, const float* y, const float* z, size_t n,
            const double T[12], float d_max, int k, int32_t* idx, float* d2o, int32_t* count)
{
    const float r2 = d_max * d_max;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        double p[3];
        xform(T, x[i], y[i], z[i], p);
        const float qx = (float)p[0], qy = (float)p[1], qz = (float)p[2];
        const int cx = cell_coord(qx, m->o[0], m->inv_h, m->dims[0]);
        const int cy = cell_coord(qy, m->o[1], m->inv_h, m->dims[1]);
        const int cz = cell_coord(qz, m->o[2], m->inv_h, m->dims[2]);
        float bd[VO_KMAX];
        int32_t bi[VO_KMAX];
        int cnt = 0;
        ROWS_BEGIN(m, cx, cy, cz)
        {
            for (int32_t j = j0; j < j1; ++j) {
                float dx = m->x[j] - qx, dyy = m->y[j] - qy, dzz = m->z[j] - qz;
                float d2 = fmaf(dzz, dzz, fmaf(dyy, dyy, dx * dx));
                if (!(d2 <= r2)) continue;
                if (cnt == k && !(d2 < bd[k - 1])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && d2 < bd[pos - 1]) {
                    bd[pos] = bd[pos - 1];
                    bi[pos] = bi[pos - 1];
                    --pos;
                }
                bd[pos] = d2;
                bi[pos] = j;
                if (cnt < k) ++cnt;
            }
        }
        ROWS_END
        for (int t = 0; t < k; ++t) {
            idx[(size_t)i * k + t] = t < cnt ? bi[t] : -1;
            d2o[(size_t)i * k + t] = t < cnt ? bd[t] : INFINITY;
        }
        if (count) count[i] = cnt;
    }
}

static inline void accum_pair(const vo_map* m, const double p[3], int32_t j, double acc[29])
{
    const double nx = m->nx[j], ny = m->ny[j], nz = m->nz[j];
    if (nx == 0.0 && ny == 0.0 && nz == 0.0) return; /* invalid normal */
    const double dx = p[0] - (double)m->x[j], dy = p[1] - (double)m->y[j],
                 dz = p[2] - (double)m->z[j];
    const double r = fma(nx, dx, fma(ny, dy, nz * dz));
    double J[6];
    J[0] = fma(p[1], nz, -(p[2] * ny));
    J[1] = fma(p[2], nx, -(p[0] * nz));
    J[2] = fma(p[0], ny, -(p[1] * nx));
    J[3] = nx;
    J[4] = ny;
    J[5] = nz;
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b, ++k) acc[k] = fma(J[a], J[b], acc[k]);
    for (int a = 0; a < 6; ++a) acc[21 + a] = fma(J[a], r, acc[21 + a]);
    acc[27] = fma(r, r, acc[27]);
    acc[28] += 1.0;
}

void vo_accumulate(const vo_map* m, const float* x, const float* y, const float* z, size_t n,
                   const double T[12], const int32_t* corr, double acc[29])
{
    memset(acc, 0, 29 * sizeof(double));
    for (size_t i = 0; i < n; ++i) {
        if (corr[i] < 0) continue;
        double p[3];
        xform(T, x[i], y[i], z[i], p);
        accum_pair(m, p, corr[i], acc);
    }
}

/* ---- a12 ------------------------------------------------------------------ */
static int ldlt6(const double Hin[36], const double b[6], double xs[6])
{
    double L[36], D[6];
    memset(L, 0, sizeof L);
    for (int j = 0; j < 6; ++j) {
        double d = Hin[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(d > 0.0)) return 1;
        D[j] = d;
        L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; ++i) {
            double v = Hin[6 * i + j];
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = v / d;
        }
    }
    double yv[6];
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * yv[k];
        yv[i] = v;
    }
    for (int i = 0; i < 6; ++i) yv[i] /= D[i];
    for (int i = 5; i >= 0; --i) {
        double v = yv[i];
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * xs[k];
        xs[i] = v;
    }
    return 0;
}

static void se3_exp_apply(const double xi[6], double T[12])
{
    const double wx = xi[0], wy = xi[1], wz = xi[2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double A, B, C;
    if (th2 < 1e-16) {
        A = 1.0 - th2 / 6.0;
        B = 0.5 - th2 / 24.0;
        C = 1.0 / 6.0 - th2 / 120.0;
    } else {
        const double th = sqrt(th2);
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
        C = (1.0 - A) / th2;
    }
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    double Rd[9], Vm[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        Rd[i] = I + A * K[i] + B * K2[i];
        Vm[i] = I + B * K[i] + C * K2[i];
    }
    double td[3];
    for (int i = 0; i < 3; ++i) td[i] = Vm[3 * i] * xi[3] + Vm[3 * i + 1] * xi[4] + Vm[3 * i + 2] * xi[5];
    double N[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j)
            N[4 * i + j] = Rd[3 * i] * T[j] + Rd[3 * i + 1] * T[4 + j] + Rd[3 * i + 2] * T[8 + j];
        N[4 * i + 3] += td[i];
    }
    memcpy(T, N, sizeof N);
}

int vo_solve_update(const double acc[29], double T[12], double xi[6])
{
    memset(xi, 0, 6 * sizeof(double));
    if (acc[28] < 6.0) return 2;
    double H[36], b[6];
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c, ++k) H[6 * a + c] = H[6 * c + a] = acc[k];
    for (int a = 0; a < 6; ++a) b[a] = -acc[21 + a];
    int rc = 0;
    if (ldlt6(H, b, xi)) {
        rc = 1;
        for (int a = 0; a < 6; ++a) H[7 * a] += 1e-9;
        if (ldlt6(H, b, xi)) {
            memset(xi, 0, 6 * sizeof(double));
            return 2;
        }
    }
    se3_exp_apply(xi, T);
    return rc;
}

int vo_icp(const vo_map* m, const float* x, const float* y, const float* z, size_t n,
           const double T0[12], int iters, float d_max, double T_out[12], vo_icp_stat* stats,
           double* trace, int threads)
{
    if (!(d_max <= m->h)) return -1;
    double T[12];
    memcpy(T, T0, sizeof T);
    const float dmax2 = d_max * d_max;
    int nth = 1;
#ifdef _OPENMP
    nth = threads > 1 ? threads : 1;
#else
    (void)threads;
#endif
    /* threads == 1: one accumulation chain over the whole frame (what the golden fixtures pin).
     * threads > 1: fixed chunks of VO_CHUNK queries handed out DYNAMICALLY (the work per query varies
     * several-fold between ground and walls: equal static shares left most threads waiting for the slowest
     * -- 2.1 x on 128 threads), one row of partial sums per CHUNK, added in chunk order:
     * the result does not depend on the thread count or on the schedule. */
    enum { VO_CHUNK = 256 };
    const size_t nchunk = nth > 1 ? (n + VO_CHUNK - 1) / VO_CHUNK : 1;
    double* part = (double*)malloc(nchunk * 32 * sizeof(double));
    uint64_t* cand = (uint64_t*)malloc(nchunk * sizeof(uint64_t));
    for (int it = 0; it < iters; ++it) {
        memset(part, 0, nchunk * 32 * sizeof(double));
        memset(cand, 0, nchunk * sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth)
        for (long long c = 0; c < (long long)nchunk; ++c) {
            const size_t lo = nth > 1 ? (size_t)c * VO_CHUNK : 0;
            const size_t hi = nth > 1 ? (lo + VO_CHUNK < n ? lo + VO_CHUNK : n) : n;
            double* acc = part + 32 * (size_t)c;
            uint64_t sc = 0;
            for (size_t i = lo; i < hi; ++i) {
                double p[3];
                xform(T, x[i], y[i], z[i], p);
                float bd;
                int32_t j = nearest(m, (float)p[0], (float)p[1], (float)p[2], &bd, &sc);
                if (j >= 0 && bd <= dmax2) accum_pair(m, p, j, acc);
            }
            cand[c] = sc;
        }
        double acc[29];
        memset(acc, 0, sizeof acc);
        uint64_t sc = 0;
        for (size_t t = 0; t < nchunk; ++t) { /* fixed chunk order */
            for (int k = 0; k < 29; ++k) acc[k] += part[32 * t + k];
            sc += cand[t];
        }
        if (stats) {
            stats[it].n_pairs = (uint32_t)acc[28];
            stats[it].rmse = acc[28] > 0 ? sqrt(acc[27] / acc[28]) : 0.0;
            stats[it].candidates = sc;
        }
        double xi[6];
        vo_solve_update(acc, T, xi);
        if (trace) memcpy(trace + 12 * it, T, sizeof T);
    }
    free(part);
    free(cand);
    memcpy(T_out, T, sizeof T);
    return 0;
}

size_t vo_increment(const vo_map* m, const float* x, const float* y, const float* z, size_t n,
                    const double T[12], int min_count, float* ox, float* oy, float* oz)
{
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        double p[3];
        xform(T, x[i], y[i], z[i], p);
        const float qx = (float)p[0], qy = (float)p[1], qz = (float)p[2];
        const int cx = cell_coord(qx, m->o[0], m->inv_h, m->dims[0]);
        const int cy = cell_coord(qy, m->o[1], m->inv_h, m->dims[1]);
        const int cz = cell_coord(qz, m->o[2], m->inv_h, m->dims[2]);
        int occ = 0;
        if (cx >= 0 && cx < m->dims[0] && cy >= 0 && cy < m->dims[1] && cz >= 0 && cz < m->dims[2]) {
            for (int fz = cz * m->S; fz < (cz + 1) * m->S; ++fz)
                for (int fy = cy * m->S; fy < (cy + 1) * m->S; ++fy) {
                    size_t row = ((size_t)fz * m->fd[1] + fy) * m->fd[0];
                    occ += m->cell_start[row + (size_t)(cx + 1) * m->S] -
                           m->cell_start[row + (size_t)cx * m->S];
                }
        }
        if (occ < min_count) {
            if (ox) {
                ox[cnt] = qx;
                oy[cnt] = qy;
                oz[cnt] = qz;
            }
            ++cnt;
        }
    }
    return cnt;
}
