/*
 * decode.c -- oracle restatement of a8: Velodyne packet decode + per-packet
 * motion compensation + frame split, i.e. HDLParser::vsInternal's
 * processHDLPacket / processFiring / pushFiringData / splitFrame.
 * TEST INFRASTRUCTURE ONLY (see velo_oracle.h).
 *
 * PARITY UNPINNED: HDLParser.cxx needs Eigen/Boost/PCL/pcap (absent), and the
 * reference has no test or capture file that pins its output.  Restated by
 * reading HDLParser.cxx:67-108 (wire structs), :179-187 (beam LUT), :587-752,
 * :754-768, :867-897, :900-977, :980-1062; quirks are reproduced, not fixed:
 *   - the first packet of a frame is recorded twice (:999 and :1009);
 *   - after a mid-packet split the rest of that packet keeps the transform that
 *     is relative to the PREVIOUS frame's carpose (:1005 runs before the loop);
 *   - "median" azimuth step = element 6 of the 11 sorted diffs (:1021-1026);
 *   - the crop flag named pointOutsideOfBox is true INSIDE the box (:629-639).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "velo_oracle.h"

#define N_ROT 36001 /* type_defs.h:16 */
#define FIRINGS 12
#define LASERS_PER_FIRING 32

typedef struct {
    float *x, *y, *z, *in, *dist;
    uint16_t* az;
    size_t n, cap;
} beam;

typedef struct {
    beam b[64];
    int nb;
    vo_pose carpose;
    int64_t t_us;
    int skips;
    size_t n_packets;
} frame;

struct vo_decoder {
    vo_laser_corr corr[64];
    int n_lasers;
    const vo_timeline* tl;
    double* cos_lut;
    double* sin_lut;
    int lut64[64];
    frame* frames;
    size_t nframes, capframes;
    frame cur;
    int meta_inited, is_hdl64, last_az, firing_skip, split_counter, points_skip;
    int crop, crop_inside;
    double region[6];
    unsigned char laser_sel[64];
};

static void frame_init(frame* f, int nb)
{
    memset(f, 0, sizeof *f);
    f->nb = nb;
    vo_pose_init(&f->carpose);
    f->t_us = VO_TIME_INVALID;
}
static void frame_free(frame* f)
{
    for (int i = 0; i < 64; ++i) {
        free(f->b[i].x);
        free(f->b[i].y);
        free(f->b[i].z);
        free(f->b[i].in);
        free(f->b[i].dist);
        free(f->b[i].az);
    }
}
static void beam_push(beam* b, float x, float y, float z, float in, uint16_t az, float dist)
{
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 2200; /* HDL_MAX_PTS_PER_LASER reserve, :573 */
        b->x = (float*)realloc(b->x, b->cap * sizeof(float));
        b->y = (float*)realloc(b->y, b->cap * sizeof(float));
        b->z = (float*)realloc(b->z, b->cap * sizeof(float));
        b->in = (float*)realloc(b->in, b->cap * sizeof(float));
        b->dist = (float*)realloc(b->dist, b->cap * sizeof(float));
        b->az = (uint16_t*)realloc(b->az, b->cap * sizeof(uint16_t));
    }
    b->x[b->n] = x;
    b->y[b->n] = y;
    b->z[b->n] = z;
    b->in[b->n] = in;
    b->az[b->n] = az;
    b->dist[b->n] = dist;
    b->n++;
}

vo_decoder* vo_decoder_new(const vo_laser_corr corr[64], int n_lasers, const vo_timeline* tl)
{
    /* HDLParser.cxx:179-181 */
    static const int lut[64] = {38, 39, 42, 43, 32, 33, 36, 37, 40, 41, 46, 47, 50, 51, 54, 55,
                                44, 45, 48, 49, 52, 53, 58, 59, 62, 63, 34, 35, 56, 57, 60, 61,
                                6,  7,  10, 11, 0,  1,  4,  5,  8,  9,  14, 15, 18, 19, 22, 23,
                                12, 13, 16, 17, 20, 21, 26, 27, 30, 31, 2,  3,  24, 25, 28, 29};
    vo_decoder* d = (vo_decoder*)calloc(1, sizeof *d);
    memcpy(d->corr, corr, sizeof d->corr);
    memcpy(d->lut64, lut, sizeof lut);
    d->n_lasers = n_lasers;
    d->tl = tl;
    d->cos_lut = (double*)malloc(N_ROT * sizeof(double));
    d->sin_lut = (double*)malloc(N_ROT * sizeof(double));
    for (unsigned i = 0; i < N_ROT; ++i) { /* :754-768 */
        double rad = (i / 100.0) * M_PI / 180.0;
        d->cos_lut[i] = cos(rad);
        d->sin_lut[i] = sin(rad);
    }
    d->last_az = -1;
    memset(d->laser_sel, 1, sizeof d->laser_sel);
    frame_init(&d->cur, n_lasers);
    return d;
}
void vo_decoder_free(vo_decoder* d)
{
    if (!d) return;
    for (size_t i = 0; i < d->nframes; ++i) frame_free(&d->frames[i]);
    free(d->frames);
    frame_free(&d->cur);
    free(d->cos_lut);
    free(d->sin_lut);
    free(d);
}
void vo_decoder_set_crop(vo_decoder* d, int enable, int crop_inside, const double region[6])
{
    d->crop = enable;
    d->crop_inside = crop_inside;
    if (region) memcpy(d->region, region, sizeof d->region);
}
void vo_decoder_set_skip(vo_decoder* d, int s) { d->firing_skip = s; }
/* setLaserSelection (HDLParser.h:106-110; consumed at HDLParser.cxx:964, indexed by laser id) */
void vo_decoder_set_laser_selection(vo_decoder* d, const unsigned char sel[64])
{
    for (int i = 0; i < 64; ++i) d->laser_sel[i] = sel[i] ? 1 : 0;
}
/* setPointsSkip (HDLParser.h:119; consumed at HDLParser.cxx:1042) */
void vo_decoder_set_points_skip(vo_decoder* d, int s) { d->points_skip = s < 0 ? 0 : s; }
int vo_decoder_num_frames(const vo_decoder* d) { return (int)d->nframes; }

/* HDLParser.cxx:867-897 */
static void split_frame(vo_decoder* d, int force)
{
    if (d->split_counter > 0 && !force) {
        d->split_counter--;
        return;
    }
    if (d->is_hdl64) {
        beam re[64];
        for (int i = 0; i < 64; ++i) re[i] = d->cur.b[d->lut64[i]];
        memcpy(d->cur.b, re, sizeof re);
    }
    if (d->nframes == d->capframes) {
        d->capframes = d->capframes ? 2 * d->capframes : 8;
        d->frames = (frame*)realloc(d->frames, d->capframes * sizeof(frame));
    }
    d->frames[d->nframes++] = d->cur;
    frame_init(&d->cur, d->n_lasers);
    d->meta_inited = 0;
}
int vo_decoder_flush(vo_decoder* d)
{
    split_frame(d, 1);
    return (int)d->nframes;
}

/* HDLParser.cxx:587-752 */
static void push_firing(vo_decoder* d, unsigned char laser_id, unsigned short azimuth,
                        unsigned short raw_dist, unsigned char raw_int, const vo_laser_corr* c,
                        const double* M)
{
    azimuth %= 36000;
    const short intensity = raw_int;
    double cos_az, sin_az;
    if (c->azimuthCorrection == 0) {
        cos_az = d->cos_lut[azimuth];
        sin_az = d->sin_lut[azimuth];
    } else {
        double rad = (((double)azimuth / 100.0) - c->azimuthCorrection) * M_PI / 180.0;
        cos_az = cos(rad);
        sin_az = sin(rad);
    }
    double distance_m = raw_dist * 0.002 + c->distanceCorrection;
    double xy = distance_m * c->cosVertCorrection;
    double pos[3] = {xy * sin_az - c->horizontalOffsetCorrection * cos_az,
                     xy * cos_az + c->horizontalOffsetCorrection * sin_az,
                     distance_m * c->sinVertCorrection + c->verticalOffsetCorrection};
    if (d->crop) {
        int in_box = pos[0] >= d->region[0] && pos[0] <= d->region[1] && pos[1] >= d->region[2] &&
                     pos[1] <= d->region[3] && pos[2] >= d->region[4] && pos[2] <= d->region[5];
        if ((in_box && !d->crop_inside) || (!in_box && d->crop_inside)) return;
    }
    if (M) vo_transform_point(pos, M);
    if (laser_id < d->cur.nb)
        beam_push(&d->cur.b[laser_id], (float)pos[0], (float)pos[1], (float)pos[2],
                  (float)intensity, azimuth, (float)distance_m);
}

static double hdl32_adjust(int block, int dsr) { return (block * 46.08) + (dsr * 1.152); }
static double vlp16_adjust(int block, int dsr, int within)
{
    return (block * 110.592) + (dsr * 2.304) + (within * 55.296);
}

/* HDLParser.cxx:900-977 */
static void process_firing(vo_decoder* d, const unsigned char* fd, int offset, int block,
                           int azimuth_diff, const double* M)
{
    const unsigned short rot = (unsigned short)(fd[2] | (fd[3] << 8));
    for (int dsr = 0; dsr < LASERS_PER_FIRING; ++dsr) {
        unsigned char laser_id = (unsigned char)(dsr + offset);
        int within = 0;
        if (d->n_lasers == 16 && laser_id >= 16) {
            laser_id -= 16;
            within = 1;
        }
        double ts_adj = 0.0, blk0 = 0.0, nblk0 = 1.0;
        if (d->n_lasers == 32) {
            ts_adj = hdl32_adjust(block, dsr);
            nblk0 = hdl32_adjust(block + 1, 0);
            blk0 = hdl32_adjust(block, 0);
        } else if (d->n_lasers == 16) {
            ts_adj = vlp16_adjust(block, laser_id, within);
            nblk0 = vlp16_adjust(block + 1, 0, 0);
            blk0 = vlp16_adjust(block, 0, 0);
        }
        int az_adj = (int)round(azimuth_diff * ((ts_adj - blk0) / (nblk0 - blk0)));
        const unsigned char* lr = fd + 4 + 3 * dsr;
        unsigned short dist = (unsigned short)(lr[0] | (lr[1] << 8));
        if (dist != 0.0 && d->laser_sel[laser_id])
            push_firing(d, laser_id, (unsigned short)(rot + az_adj), dist, lr[2],
                        &d->corr[dsr + offset], M);
    }
}

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* HDLParser.cxx:980-1055 (+ :1057-1062) */
int vo_decoder_packet(vo_decoder* d, const unsigned char* data, size_t len, int64_t t_us)
{
    if (len != 1206) return (int)d->nframes;
    vo_pose tr;
    vo_pose_init(&tr);
    if (d->tl)
        vo_interpolate_transform(d->tl, t_us, &tr);
    else
        tr.t_us = t_us;
    if (!d->meta_inited) {
        d->cur.carpose = tr; /* memcpy, :995 */
        d->cur.t_us = t_us;
        d->cur.skips = d->firing_skip;
        d->cur.n_packets++; /* :999 */
        d->meta_inited = 1;
    }
    tr.t_us = t_us;
    double M[12];
    const double* Mp = NULL;
    if (tr.seconds_pos != -1) {
        for (int i = 0; i < 3; ++i) tr.T[i] -= d->cur.carpose.T[i]; /* :1057-1062 */
        vo_pose_matrix(&tr, M);
        Mp = M;
    }
    d->cur.n_packets++; /* :1009 */
    int block = d->firing_skip;
    d->firing_skip = 0;
    int diffs[FIRINGS - 1];
    for (int i = 0; i < FIRINGS - 1; ++i) {
        int r1 = data[100 * (i + 1) + 2] | (data[100 * (i + 1) + 3] << 8);
        int r0 = data[100 * i + 2] | (data[100 * i + 3] << 8);
        diffs[i] = (36000 + r1 - r0) % 36000;
    }
    qsort(diffs, FIRINGS - 1, sizeof(int), cmp_int);
    int azimuth_diff = diffs[FIRINGS / 2];
    for (; block < FIRINGS; ++block) {
        const unsigned char* fd = data + 100 * block;
        unsigned short id = (unsigned short)(fd[0] | (fd[1] << 8));
        int rot = fd[2] | (fd[3] << 8);
        int offset = (id == 0xeeff) ? 0 : 32;
        d->is_hdl64 |= (offset > 0);
        if (rot < d->last_az) {
            d->firing_skip = block;
            split_frame(d, 0);
        }
        if (d->points_skip == 0 || block % (d->points_skip + 1) == 0)
            process_firing(d, fd, offset, block, azimuth_diff, Mp);
        d->last_az = rot;
    }
    return (int)d->nframes;
}

size_t vo_frame_beam_size(const vo_decoder* d, int f, int b) { return d->frames[f].b[b].n; }
void vo_frame_beam_copy(const vo_decoder* d, int f, int b, float* x, float* y, float* z,
                        float* in, uint16_t* az, float* dist)
{
    const beam* k = &d->frames[f].b[b];
    if (x) memcpy(x, k->x, k->n * sizeof(float));
    if (y) memcpy(y, k->y, k->n * sizeof(float));
    if (z) memcpy(z, k->z, k->n * sizeof(float));
    if (in) memcpy(in, k->in, k->n * sizeof(float));
    if (az) memcpy(az, k->az, k->n * sizeof(uint16_t));
    if (dist) memcpy(dist, k->dist, k->n * sizeof(float));
}
void vo_frame_carpose(const vo_decoder* d, int f, vo_pose* out, int64_t* t_us, int* skips)
{
    if (out) *out = d->frames[f].carpose;
    if (t_us) *t_us = d->frames[f].t_us;
    if (skips) *skips = d->frames[f].skips;
}
size_t vo_frame_num_packets(const vo_decoder* d, int f) { return d->frames[f].n_packets; }

/* ---- calibration file, HDLParser.cxx:771-858 ------------------------------------------------
 * Restated without boost::property_tree: the reference looks elements up by name only
 * (boost_serialization.DB.enabled_ / .points_ > item > px > <field>), atoi/atof on the text.
 * PARITY UNPINNED (Boost absent here); product and oracle are two independent scanners held
 * to each other and to the numbers a generated file was written from. */
static const char* find_elem(const char* s, const char* end, const char* name, const char** content_end)
{
    const size_t nl = strlen(name);
    const char* p = s;
    while (p < end) {
        p = memchr(p, '<', (size_t)(end - p));
        if (!p) return NULL;
        if ((size_t)(end - p) > nl + 1 && strncmp(p + 1, name, nl) == 0 &&
            (p[1 + nl] == '>' || p[1 + nl] == ' ' || p[1 + nl] == '\t' || p[1 + nl] == '\n' || p[1 + nl] == '\r')) {
            const char* gt = memchr(p, '>', (size_t)(end - p));
            if (!gt) return NULL;
            /* closing tag */
            const char* q = gt + 1;
            while (q < end) {
                q = memchr(q, '<', (size_t)(end - q));
                if (!q) return NULL;
                if ((size_t)(end - q) > nl + 2 && q[1] == '/' && strncmp(q + 2, name, nl) == 0 && q[2 + nl] == '>') {
                    *content_end = q;
                    return gt + 1;
                }
                ++q;
            }
            return NULL;
        }
        ++p;
    }
    return NULL;
}

static int elem_number(const char* s, const char* end, const char* name, double* out)
{
    const char* ce;
    const char* c = find_elem(s, end, name, &ce);
    if (!c) return 0;
    char buf[64];
    size_t n = (size_t)(ce - c);
    if (n >= sizeof buf) n = sizeof buf - 1;
    memcpy(buf, c, n);
    buf[n] = 0;
    *out = atof(buf);
    return 1;
}

int vo_load_corrections(const char* path, vo_laser_corr corr[64], int* n_enabled)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)sz + 1);
    if (fread(s, 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        free(s);
        return -1;
    }
    fclose(f);
    s[sz] = 0;
    const char* end = s + sz;
    memset(corr, 0, 64 * sizeof(vo_laser_corr));
    const char *db_e, *db = find_elem(s, end, "DB", &db_e);
    if (!db) {
        free(s);
        return -1;
    }
    int enabled = 0;
    const char *en_e, *en = find_elem(db, db_e, "enabled_", &en_e);
    if (en) {
        const char* p = en;
        for (;;) {
            const char *ie, *it = find_elem(p, en_e, "item", &ie);
            if (!it) break;
            if (atoi(it) == 1 && it != ie) ++enabled;
            p = ie + 1;
        }
    }
    if (n_enabled) *n_enabled = enabled;
    const char *pt_e, *pt = find_elem(db, db_e, "points_", &pt_e);
    if (!pt) {
        free(s);
        return -1;
    }
    const char* p = pt;
    for (;;) {
        const char *xe, *x = find_elem(p, pt_e, "px", &xe);
        if (!x) break;
        p = xe + 1;
        double id = -1, az = 0, vert = 0, dist = 0, voff = 0, hoff = 0;
        const char* ce;
        const char* c = find_elem(x, xe, "id_", &ce);
        if (c) id = (double)atoi(c);
        elem_number(x, xe, "rotCorrection_", &az);
        elem_number(x, xe, "vertCorrection_", &vert);
        elem_number(x, xe, "distCorrection_", &dist);
        elem_number(x, xe, "vertOffsetCorrection_", &voff);
        elem_number(x, xe, "horizOffsetCorrection_", &hoff);
        const int index = (int)id;
        if (index < 0 || index >= 64) continue;
        vo_laser_corr* k = &corr[index];
        k->azimuthCorrection = az;
        k->verticalCorrection = vert;
        k->distanceCorrection = dist / 100.0;          /* :836 */
        k->verticalOffsetCorrection = voff / 100.0;    /* :837 */
        k->horizontalOffsetCorrection = hoff / 100.0;  /* :838 */
        k->cosVertCorrection = cos(k->verticalCorrection * M_PI / 180.0); /* :840 */
        k->sinVertCorrection = sin(k->verticalCorrection * M_PI / 180.0); /* :841 */
    }
    for (int i = 0; i < 64; ++i) { /* :848-855 */
        corr[i].sinVertOffsetCorrection = corr[i].verticalOffsetCorrection * corr[i].sinVertCorrection;
        corr[i].cosVertOffsetCorrection = corr[i].verticalOffsetCorrection * corr[i].cosVertCorrection;
    }
    free(s);
    return 0;
}
