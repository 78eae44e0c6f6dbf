/*
 * pose.c -- oracle restatement of a3..a7: PoseTransform algebra, Euler->matrix,
 * the per-point affine transform, TimeLine<PoseTransform> and
 * TransformManager::interpolateTransform.
 * TEST INFRASTRUCTURE ONLY (see velo_oracle.h).
 *
 * PARITY UNPINNED: type_defs.h, TimeLine.h and TransformManager.cxx include
 * Eigen / Boost / glog, which are neither installed nor vendored, so the
 * reference cannot be compiled here and it ships no test that pins a result.
 * What follows is a restatement by reading, property-tested in
 * tests/test_oracle_pose.py (identity, composition cross-check against scipy's
 * intrinsic 'YXZ', knot/extrapolation behaviour).
 *
 * Third-party algorithm restated (dependency absent from /root/reference):
 *   Eigen 3.x (CMakeLists.txt:78, version unpinned)
 *     AngleAxis<double>::toRotationMatrix():
 *        sin_axis = sin(a)*axis; c = cos(a); cos1_axis = (1-c)*axis;
 *        off-diagonals tmp +- sin_axis component; diag = cos1_axis*axis + c
 *     Transform::rotate(R): linear = linear * R   (right-multiply)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "velo_oracle.h"

/* type_defs.cxx:47-57 */
void vo_pose_init(vo_pose* p)
{
    for (int i = 0; i < 3; ++i) p->T[i] = p->R[i] = p->V[i] = 0;
    p->t_us = VO_TIME_INVALID;
    p->week_number = 0;
    p->milliseconds = p->week_number_pos = 0;
    p->seconds_pos = -1;
}

/* type_defs.h:102-131: operators touch T,R,V only; everything else in the
 * result is default-constructed (timestamp invalid, seconds_pos = -1). */
void vo_pose_add(const vo_pose* a, const vo_pose* b, vo_pose* o)
{
    vo_pose r;
    vo_pose_init(&r);
    for (int i = 0; i < 3; ++i) {
        r.T[i] = a->T[i] + b->T[i];
        r.R[i] = a->R[i] + b->R[i];
        r.V[i] = a->V[i] + b->V[i];
    }
    *o = r;
}
void vo_pose_sub(const vo_pose* a, const vo_pose* b, vo_pose* o)
{
    vo_pose r;
    vo_pose_init(&r);
    for (int i = 0; i < 3; ++i) {
        r.T[i] = a->T[i] - b->T[i];
        r.R[i] = a->R[i] - b->R[i];
        r.V[i] = a->V[i] - b->V[i];
    }
    *o = r;
}
void vo_pose_scale(const vo_pose* a, double ratio, vo_pose* o)
{
    vo_pose r;
    vo_pose_init(&r);
    for (int i = 0; i < 3; ++i) {
        r.T[i] = a->T[i] * ratio;
        r.R[i] = a->R[i] * ratio;
        r.V[i] = a->V[i] * ratio;
    }
    *o = r;
}

/* Eigen AngleAxis::toRotationMatrix, restated (see header comment). */
static void angle_axis_matrix(double angle, const double ax[3], double R[9])
{
    double s, c;
    sincos(angle, &s, &c); /* one libm entry point on both sides of the parity check */
    const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const double c1[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
    double tmp;
    tmp = c1[0] * ax[1];
    R[1] = tmp - sa[2];
    R[3] = tmp + sa[2];
    tmp = c1[0] * ax[2];
    R[2] = tmp + sa[1];
    R[6] = tmp - sa[1];
    tmp = c1[1] * ax[2];
    R[5] = tmp - sa[0];
    R[7] = tmp + sa[0];
    R[0] = c1[0] * ax[0] + c;
    R[4] = c1[1] * ax[1] + c;
    R[8] = c1[2] * ax[2] + c;
}

static void mat3_rmul(double L[9], const double R[9])
{
    double o[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[3 * i + j] = (L[3 * i] * R[j] + L[3 * i + 1] * R[3 + j]) + L[3 * i + 2] * R[6 + j];
    memcpy(L, o, sizeof o);
}

#define VO_TO_RADIUS(deg) ((deg)*M_PI / 180) /* type_defs.h:25 */

/* type_defs.h:134-146: I.rotate(Y, R[0]).rotate(X, R[1]).rotate(Z, R[2]); translation = T.
 * M is row-major 3x4. */
void vo_pose_matrix(const vo_pose* p, double M[12])
{
    static const double UY[3] = {0, 1, 0}, UX[3] = {1, 0, 0}, UZ[3] = {0, 0, 1};
    double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, R[9];
    angle_axis_matrix(VO_TO_RADIUS(p->R[0]), UY, R);
    mat3_rmul(L, R);
    angle_axis_matrix(VO_TO_RADIUS(p->R[1]), UX, R);
    mat3_rmul(L, R);
    angle_axis_matrix(VO_TO_RADIUS(p->R[2]), UZ, R);
    mat3_rmul(L, R);
    for (int i = 0; i < 3; ++i) {
        M[4 * i + 0] = L[3 * i + 0];
        M[4 * i + 1] = L[3 * i + 1];
        M[4 * i + 2] = L[3 * i + 2];
        M[4 * i + 3] = p->T[i];
    }
}

/* Inverse of a5 for reporting: linear = Ry(a) Rx(b) Rz(c)
 *   M12 = -sin b ; M02 = sin a cos b ; M22 = cos a cos b ; M10 = cos b sin c ; M11 = cos b cos c */
void vo_matrix_to_TRdeg(const double M[12], double TRdeg[6])
{
    TRdeg[0] = M[3];
    TRdeg[1] = M[7];
    TRdeg[2] = M[11];
    double sb = -M[6];
    if (sb > 1) sb = 1;
    if (sb < -1) sb = -1;
    TRdeg[3] = atan2(M[2], M[10]) * 180 / M_PI;
    TRdeg[4] = asin(sb) * 180 / M_PI;
    TRdeg[5] = atan2(M[4], M[5]) * 180 / M_PI;
}

/* type_defs.h:160-166: ((m0*x + m1*y) + m2*z) + m3, products rounded separately. */
void vo_transform_point(double pt[3], const double M[12])
{
    const double x = pt[0], y = pt[1], z = pt[2];
    pt[0] = M[0] * x + M[1] * y + M[2] * z + M[3];
    pt[1] = M[4] * x + M[5] * y + M[6] * z + M[7];
    pt[2] = M[8] * x + M[9] * y + M[10] * z + M[11];
}

/* a7 as a batch (the thing K1 is checked against): the float point is widened
 * to double, transformed as above, and rounded once to float
 * (HDLParser.cxx:731-736). */
void vo_compensate(const float* x, const float* y, const float* z, const uint16_t* pkt, size_t n,
                   const double* T3x4, size_t n_pkt, float* ox, float* oy, float* oz)
{
    for (size_t i = 0; i < n; ++i) {
        size_t k = pkt[i];
        if (k >= n_pkt) k = n_pkt - 1;
        double p[3] = {x[i], y[i], z[i]};
        vo_transform_point(p, T3x4 + 12 * k);
        ox[i] = (float)p[0];
        oy[i] = (float)p[1];
        oz[i] = (float)p[2];
    }
}

/* ------------------------------------------------------------------------- */
/* TimeLine<PoseTransform>: vector<vector<shared_ptr>> buckets + 5-slot ring. */

typedef struct {
    vo_pose* v;
    size_t n, cap;
} bucket;

struct vo_timeline {
    bucket* b; /* timeline */
    size_t nb, capb;
    vo_pose ring[5]; /* boost::circular_buffer<...>(5), kept in logical order */
    int nring;
    int64_t start_us, max_us;
    double interval;
    int finalized;
    size_t total;
};

static void bucket_insert(bucket* k, size_t pos, const vo_pose* p)
{
    if (k->n == k->cap) {
        k->cap = k->cap ? 2 * k->cap : 2;
        k->v = (vo_pose*)realloc(k->v, k->cap * sizeof(vo_pose));
    }
    memmove(k->v + pos + 1, k->v + pos, (k->n - pos) * sizeof(vo_pose));
    k->v[pos] = *p;
    k->n++;
}
static void tl_insert_bucket(vo_timeline* t, size_t pos)
{
    if (t->nb == t->capb) {
        t->capb = t->capb ? 2 * t->capb : 16;
        t->b = (bucket*)realloc(t->b, t->capb * sizeof(bucket));
    }
    memmove(t->b + pos + 1, t->b + pos, (t->nb - pos) * sizeof(bucket));
    memset(t->b + pos, 0, sizeof(bucket));
    t->nb++;
}
static void ring_push_back(vo_timeline* t, const vo_pose* p)
{
    if (t->nring == 5) {
        memmove(t->ring, t->ring + 1, 4 * sizeof(vo_pose));
        t->ring[4] = *p;
    } else
        t->ring[t->nring++] = *p;
}
static void ring_push_front(vo_timeline* t, const vo_pose* p)
{
    if (t->nring < 5) t->nring++;
    memmove(t->ring + 1, t->ring, (size_t)(t->nring - 1) * sizeof(vo_pose));
    t->ring[0] = *p;
}
/* circular_buffer::insert(pos,item): when full, the first element is dropped;
 * inserting at begin() of a full buffer is a no-op. */
static void ring_insert(vo_timeline* t, int pos, const vo_pose* p)
{
    if (t->nring == 5) {
        if (pos == 0) return;
        memmove(t->ring, t->ring + 1, (size_t)(pos - 1) * sizeof(vo_pose));
        t->ring[pos - 1] = *p;
    } else {
        memmove(t->ring + pos + 1, t->ring + pos, (size_t)(t->nring - pos) * sizeof(vo_pose));
        t->ring[pos] = *p;
        t->nring++;
    }
}

vo_timeline* vo_timeline_new(void)
{
    vo_timeline* t = (vo_timeline*)calloc(1, sizeof *t);
    t->start_us = t->max_us = VO_TIME_INVALID;
    return t;
}
void vo_timeline_free(vo_timeline* t)
{
    if (!t) return;
    for (size_t i = 0; i < t->nb; ++i) free(t->b[i].v);
    free(t->b);
    free(t);
}
size_t vo_timeline_size(const vo_timeline* t) { return t->total; }

/* TimeLine.h:536-552 */
static void tl_rearrange(vo_timeline* t)
{
    /* long / size_t: unsigned integer division, then stored in a double */
    t->interval = (double)((uint64_t)(t->max_us - t->start_us) / (uint64_t)t->total);
    size_t n = 0;
    vo_pose* all = (vo_pose*)malloc(t->total * sizeof(vo_pose) + 1);
    for (size_t i = 0; i < t->nb; ++i) {
        for (size_t j = 0; j < t->b[i].n; ++j) all[n++] = t->b[i].v[j];
        free(t->b[i].v);
    }
    t->nb = 0;
    t->nring = 0;
    for (size_t i = 0; i < n; ++i) {
        int index = (int)((double)(all[i].t_us - t->start_us) / t->interval);
        while ((long)t->nb <= (long)index) tl_insert_bucket(t, t->nb);
        bucket_insert(&t->b[index], t->b[index].n, &all[i]);
        ring_push_back(t, &all[i]);
    }
    free(all);
    t->finalized = 1;
}

/* TimeLine.h:140-226 */
void vo_timeline_add(vo_timeline* t, const vo_pose* p)
{
    const int64_t ts = p->t_us;
    if (t->nb == 0) {
        t->start_us = t->max_us = ts;
        tl_insert_bucket(t, 0);
        bucket_insert(&t->b[0], 0, p);
        ring_push_back(t, p);
        t->total++;
    } else if (t->nb == 1) {
        if (ts == t->start_us) { /* overwrite */
            t->b[0].v[t->b[0].n - 1] = *p;
            t->ring[0] = *p;
        } else if (ts < t->start_us) {
            t->interval = (double)(t->start_us - ts) * 0.95;
            tl_insert_bucket(t, 0);
            bucket_insert(&t->b[0], 0, p);
            ring_push_front(t, p);
            t->max_us = t->start_us;
            t->start_us = ts;
            t->total++;
        } else {
            t->interval = (double)(ts - t->start_us) * 0.95;
            tl_insert_bucket(t, t->nb);
            bucket_insert(&t->b[t->nb - 1], 0, p);
            ring_push_back(t, p);
            t->max_us = ts;
            t->total++;
        }
    } else {
        if (!t->finalized && t->total == 10) tl_rearrange(t);
        if (ts > t->ring[t->nring - 1].t_us) {
            ring_push_back(t, p);
        } else {
            int c = 0;
            while (t->ring[c].t_us < ts) ++c;
            if (t->ring[c].t_us == ts)
                t->ring[c] = *p;
            else
                ring_insert(t, c, p);
        }
        int index = (int)((double)(ts - t->start_us) / t->interval);
        if (ts >= t->start_us) {
            while ((long)t->nb <= (long)index) tl_insert_bucket(t, t->nb);
            bucket* k = &t->b[index];
            size_t pos = 0;
            while (pos < k->n && k->v[pos].t_us < ts) ++pos;
            if (pos == k->n) {
                bucket_insert(k, pos, p);
                t->max_us = ts; /* :195 -- set even for an out-of-order insert */
                t->total++;
            } else if (k->v[pos].t_us == ts) {
                k->v[pos] = *p;
            } else {
                bucket_insert(k, pos, p);
                t->total++;
            }
        } else {
            index = (int)floor((double)(ts - t->start_us) / t->interval);
            while ((index++) != 0) tl_insert_bucket(t, 0);
            bucket_insert(&t->b[0], t->b[0].n, p);
            t->start_us = ts;
            t->total++;
        }
    }
}

/* TimeLine.h:384-468.  Returns how many of (fore, back) are valid.  Where the
 * reference would index out of range (undefined behaviour, e.g. an empty
 * bucket 1 at :396) this returns what is defined so far. */
int vo_timeline_boundary(const vo_timeline* t, int64_t q, vo_pose* fore, vo_pose* back)
{
    if (t->nb == 0) return 0;
    if (t->nb == 1) {
        *fore = t->b[0].v[0];
        return 1;
    }
    if (q <= t->start_us) {
        *fore = t->b[0].v[0];
        if (t->b[0].n > 1)
            *back = t->b[0].v[1];
        else if (t->b[1].n > 0)
            *back = t->b[1].v[0];
        else
            return 1;
        return 2;
    }
    if (q >= t->max_us) {
        *back = t->ring[t->nring - 1];
        *fore = t->ring[t->nring - 2];
        return 2;
    }
    if (q > t->ring[0].t_us) {
        int i = 1;
        while (i < t->nring - 1 && t->ring[i].t_us < q) ++i;
        *fore = t->ring[i - 1];
        *back = t->ring[i];
        return 2;
    }
    long index = (long)(int)((double)(q - t->start_us) / t->interval);
    int have_back = 0;
    if (t->b[index].n != 0) {
        const bucket* k = &t->b[index];
        if (k->v[0].t_us <= q) {
            *fore = k->v[0];
            for (size_t i = 1; i < k->n; ++i) {
                if (k->v[i].t_us < q)
                    *fore = k->v[i];
                else {
                    *back = k->v[i];
                    have_back = 1;
                    break;
                }
            }
            if (!have_back) {
                size_t c = (size_t)index + 1;
                while (c != t->nb && t->b[c].n == 0) ++c;
                if (c != t->nb) {
                    *back = t->b[c].v[0];
                    have_back = 1;
                }
            }
            if (k->v[0].t_us == q) { /* :426-444 */
                long c = index - 1;
                while (c >= 0 && t->b[c].n == 0) --c;
                if (c != -1) {
                    const vo_pose* other = &t->b[c].v[t->b[c].n - 1];
                    if (have_back) {
                        int64_t diff_f = q - back->t_us;
                        int64_t diff_b = other->t_us - q;
                        if (diff_f > diff_b) {
                            *back = *fore;
                            *fore = *other;
                        }
                    } else {
                        *back = *fore;
                        *fore = *other;
                        have_back = 1;
                    }
                }
            }
        } else {
            *back = k->v[0];
            have_back = 1;
            long c = index - 1;
            while (c >= 0 && t->b[c].n == 0) --c;
            *fore = t->b[c].v[t->b[c].n - 1];
        }
    } else {
        long c = index - 1;
        while (c >= 0 && t->b[c].n == 0) --c;
        *fore = t->b[c].v[t->b[c].n - 1];
        while (t->b[++index].n == 0) {
        }
        *back = t->b[index].v[0];
        have_back = 1;
    }
    return have_back ? 2 : 1;
}

/* TransformManager.cxx:149-177 */
int vo_interpolate_transform(const vo_timeline* t, int64_t q, vo_pose* out)
{
    vo_pose fore, back;
    out->t_us = q; /* :151 */
    int nb = vo_timeline_boundary(t, q, &fore, &back);
    if (nb == 0) return 0;
    if (nb == 1) {
        /* :161: long / 1e6f is a FLOAT division, widened afterwards */
        double sec = (double)((float)(q - fore.t_us) / 1e6f);
        for (int i = 0; i < 3; ++i) {
            out->V[i] = fore.V[i];
            out->R[i] = fore.R[i];
            out->T[i] = fore.T[i] + fore.V[i] * sec;
        }
        return 1; /* seconds_pos is left untouched (-1 on a fresh pose) */
    }
    double ratio = (double)(q - fore.t_us) / (double)(back.t_us - fore.t_us);
    vo_pose d, s, r;
    vo_pose_sub(&back, &fore, &d);
    vo_pose_scale(&d, ratio, &s);
    vo_pose_add(&fore, &s, &r);
    *out = r;             /* :173: whole-struct assignment; timestamp becomes invalid */
    out->seconds_pos = 0; /* :174 */
    return 1;
}

/* ---- ptimeToWeekMilli, type_defs.cxx:74-79 --------------------------------------------------
 *   week  = t.date().week_number();
 *   beginOfWeek = date(t.date() - days(to_tm(t.date()).tm_wday));   (00:00 of the last Sunday)
 *   milli = (t - beginOfWeek).total_milliseconds();                 (into a uint32)
 * boost::gregorian::date::week_number lives in Boost.DateTime (a dependency the reference does not
 * vendor or pin: CMakeLists.txt:98 `find_package(Boost ... date_time ...)`, absent from this image);
 * its published algorithm (boost/date_time/gregorian_calendar.ipp, gregorian_calendar_base::
 * week_number) is restated below on julian day numbers.  Unpinned to the reference's own object
 * code; pinned to the ISO 8601 definition by tests/test_host_parity.py (== Python's
 * date.isocalendar() on every day of 1970-2199). */
static long vo_jdn(long y, long m, long d) /* gregorian_calendar.ipp: julian_day_number */
{
    const long a = (14 - m) / 12;
    const long yy = y + 4800 - a;
    const long mm = m + 12 * a - 3;
    return d + (153 * mm + 2) / 5 + 365 * yy + yy / 4 - yy / 100 + yy / 400 - 32045;
}
static void vo_civil(long jdn, long* y, long* m, long* d) /* gregorian_calendar.ipp: from_day_number */
{
    const long a = jdn + 32044;
    const long b = (4 * a + 3) / 146097;
    const long c = a - (146097 * b) / 4;
    const long dd = (4 * c + 3) / 1461;
    const long e = c - (1461 * dd) / 4;
    const long mm = (5 * e + 2) / 153;
    *d = e - (153 * mm + 2) / 5 + 1;
    *m = mm + 3 - 12 * (mm / 10);
    *y = 100 * b + dd - 4800 + mm / 10;
}
void vo_time_to_week_milli(int64_t t_us, uint16_t* week, uint32_t* milli)
{
    const int64_t day_us = 86400LL * 1000000;
    int64_t days = t_us / day_us;
    if (t_us % day_us < 0) --days;
    const long today = 2440588L + (long)days; /* JDN of 1970-01-01 */
    const long wday = (today + 1) % 7;        /* tm_wday: 0 = Sunday */
    *milli = (uint32_t)((t_us - (days - wday) * day_us) / 1000);
    long y, m, d;
    vo_civil(today, &y, &m, &d);
    unsigned long begin = (unsigned long)vo_jdn(y, 1, 1);
    unsigned long day = (begin + 3) % 7;
    unsigned long w = ((unsigned long)today + day - begin + 4) / 7;
    if (w >= 1 && w <= 52) {
        *week = (uint16_t)w;
        return;
    }
    if (w == 53) {
        const int leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
        *week = (day == 6 || (day == 5 && leap)) ? 53 : 1;
        return;
    }
    /* week 0: the date belongs to the last week of the previous year */
    begin = (unsigned long)vo_jdn(y - 1, 1, 1);
    day = (begin + 3) % 7;
    *week = (uint16_t)(((unsigned long)today + day - begin + 4) / 7);
}
