// Kernels whose outputs feed INTEGER decisions (positive/negative ROI partition, argmax class,
// no-object mask) and therefore must agree bit-for-bit with the CPU restatement:
//   yolo_decode / yolo_detections  (DecodeYOLOLayer model.py:1442-1473, DetectionsLayer :1493-1538)
//   mask_targets                   (detect_mask_target_graph model.py:457-602 + helpers)
//   yolo_loss fwd+bwd              (yolo_custom_loss model.py:86-242)
// This translation unit is compiled with -ffp-contract=off: every float expression below is a
// sequence of individually rounded IEEE-754 binary32 operations in source order, and exp/sigmoid
// are the explicit polynomial of exact_math.h (same operation sequence as the oracle's det_expf).
#include "myolo_common.h"
#include "exact_math.h"

// ---------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_box(const float* __restrict__ p, const float* __restrict__ anchors, int a, int col,
                                           int row, float Gf, float* out4)
{
    const float x = (myolo_sigmoidf(p[0]) + (float)col) / Gf;
    const float y = (myolo_sigmoidf(p[1]) + (float)row) / Gf;
    const float w = (myolo_expf(p[2]) * anchors[2 * a + 0]) / Gf;
    const float h = (myolo_expf(p[3]) * anchors[2 * a + 1]) / Gf;
    const float hw = w / 2.0f, hh = h / 2.0f;
    out4[0] = x - hw;
    out4[1] = y - hh;
    out4[2] = x + hw;
    out4[3] = y + hh;
}

__global__ void yolo_decode_kernel(const float* __restrict__ yp, const float* __restrict__ anchors, float* __restrict__ out,
                                   int total, int G, int A, int D, int det)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int a = i % A;
    const int col = (i / A) % G;
    const int row = (i / (A * G)) % G;
    const float* p = yp + (long long)i * D;
    float b[4];
    decode_box(p, anchors, a, col, row, (float)G, b);
    if (!det) {
        float* o = out + (long long)i * 4;
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
    } else {
        float* o = out + (long long)i * 6;
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
        o[4] = myolo_sigmoidf(p[4]);
        int best = 0;
        float bv = p[5];
        for (int k = 1; k < D - 5; ++k)
            if (p[5 + k] > bv) { bv = p[5 + k]; best = k; }     // first maximum wins (tf.argmax)
        o[5] = (float)best;
    }
}

// ---------------------------------------------------------------------------------------
// mask targets: one block per image
// ---------------------------------------------------------------------------------------
#define MT_MAXT 64
#define MT_MAXR 2048
__global__ __launch_bounds__(256) void mask_targets_kernel(const float* __restrict__ proposals, const int32_t* __restrict__ gt_ids,
                                                           const int32_t* __restrict__ gt_boxes, const uint8_t* __restrict__ gt_masks,
                                                           float* __restrict__ rois, int32_t* __restrict__ tcls,
                                                           float* __restrict__ tmasks, int32_t* __restrict__ npos_out,
                                                           int R, int T, int H, int W, int mh, int mw)
{
    __shared__ float gtb[MT_MAXT][4];
    __shared__ int keep[MT_MAXT];
    __shared__ int nkeep_s;
    __shared__ unsigned char posf[MT_MAXR];
    __shared__ short arg[MT_MAXR];
    __shared__ short dest[MT_MAXR];
    __shared__ short src_of_dest[MT_MAXR];
    __shared__ int npos_s;
    __shared__ int wcnt[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* prop = proposals + (long long)b * R * 4;

    // norm_boxes_graph (model.py:1405-1408): (box - [0,0,1,1]) / ([H,W,H,W] - 1)   (reference splits (H,W) as w,h)
    if (tid < T) {
        const int32_t* bp = gt_boxes + ((long long)b * T + tid) * 4;
        gtb[tid][0] = ((float)bp[0] - 0.0f) / (float)(H - 1);
        gtb[tid][1] = ((float)bp[1] - 0.0f) / (float)(W - 1);
        gtb[tid][2] = ((float)bp[2] - 1.0f) / (float)(H - 1);
        gtb[tid][3] = ((float)bp[3] - 1.0f) / (float)(W - 1);
    }
    __syncthreads();
    if (tid == 0) {   // trim_zeros_graph (model.py:1418): keep rows whose |.| sum is non-zero
        int n = 0;
        for (int g = 0; g < T; ++g) {
            const float s = fabsf(gtb[g][0]) + fabsf(gtb[g][1]) + fabsf(gtb[g][2]) + fabsf(gtb[g][3]);
            if (s != 0.0f) keep[n++] = g;
        }
        nkeep_s = n;
    }
    __syncthreads();
    const int nkeep = nkeep_s;
    // overlaps_graph (model.py:438-451) + reduce_max / argmax over the kept GT boxes
    for (int r = tid; r < R; r += blockDim.x) {
        const float bx1 = prop[r * 4 + 0], by1 = prop[r * 4 + 1], bx2 = prop[r * 4 + 2], by2 = prop[r * 4 + 3];
        const float a1 = (by2 - by1) * (bx2 - bx1);
        float best = -INFINITY;
        int bi = 0;
        for (int k = 0; k < nkeep; ++k) {
            const int g = keep[k];
            const float x1 = fmaxf(bx1, gtb[g][0]), y1 = fmaxf(by1, gtb[g][1]);
            const float x2 = fminf(bx2, gtb[g][2]), y2 = fminf(by2, gtb[g][3]);
            const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
            const float a2 = (gtb[g][3] - gtb[g][1]) * (gtb[g][2] - gtb[g][0]);
            const float uni = a1 + a2 - inter;
            const float iou = inter / uni;
            if (iou > best) { best = iou; bi = k; }
        }
        // positive: max IoU >= 0.5 ; negative: < 0.5 ; NaN belongs to neither (2 = dropped)
        posf[r] = (best >= 0.5f) ? 1 : ((best < 0.5f) ? 0 : 2);
        arg[r] = (short)bi;
    }
    __syncthreads();
    // stable partition: positives first, then negatives, index order kept -- ranks by ballots over chunks of 256 proposals (a single thread walking
    // the R proposals four times through LDS took ~15 us of the step's critical path)
    {
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long lower = (1ull << lane) - 1ull;
        int base_p = 0, base_n = 0;
        for (int r0 = 0; r0 < R; r0 += 256) {
            const int r = r0 + tid;
            const int f = r < R ? posf[r] : 2;
            const unsigned long long bp = __ballot(f == 1), bn = __ballot(f == 0);
            if (lane == 0) { wcnt[0][wave] = __popcll(bp); wcnt[1][wave] = __popcll(bn); }
            __syncthreads();
            int offp = base_p, offn = base_n;
            for (int w2 = 0; w2 < wave; ++w2) { offp += wcnt[0][w2]; offn += wcnt[1][w2]; }
            if (f == 1) dest[r] = (short)(offp + __popcll(bp & lower));
            else if (f == 0) dest[r] = (short)(-2 - (offn + __popcll(bn & lower)));         // negatives: rank among negatives, shifted below
            else if (r < R) dest[r] = -1;
            base_p += wcnt[0][0] + wcnt[0][1] + wcnt[0][2] + wcnt[0][3];
            base_n += wcnt[1][0] + wcnt[1][1] + wcnt[1][2] + wcnt[1][3];
            __syncthreads();
        }
        if (tid == 0) { npos_s = base_p; if (blockIdx.y == 0) npos_out[b] = base_p; }
        for (int d = tid; d < R; d += blockDim.x) src_of_dest[d] = -1;
        __syncthreads();
        for (int r = tid; r < R; r += blockDim.x) {
            int d = dest[r];
            if (d <= -2) { d = base_p + (-2 - d); dest[r] = (short)d; }
            if (d >= 0) src_of_dest[d] = (short)r;
        }
    }
    __syncthreads();
    const int npos = npos_s;
    // rois + class ids
    // (every workgroup of an image repeats the matching above -- a few microseconds -- and fills its share of the image's target masks, 115 000
    // values at 28 x 28 and R = 147; the first one also writes the ROIs and class ids)
    if (blockIdx.y == 0)
    for (int d = tid; d < R; d += blockDim.x) {
        const int r = src_of_dest[d];
        float* o = rois + ((long long)b * R + d) * 4;
        if (r >= 0) {
            o[0] = prop[r * 4 + 0]; o[1] = prop[r * 4 + 1]; o[2] = prop[r * 4 + 2]; o[3] = prop[r * 4 + 3];
        } else {
            o[0] = o[1] = o[2] = o[3] = 0.0f;
        }
        int cls = 0;
        if (r >= 0 && d < npos) cls = gt_ids[(long long)b * T + keep[arg[r]]];
        tcls[(long long)b * R + d] = cls;
    }
    // mask targets: crop_and_resize(gt_mask[g], [y1,x1,y2,x2], mh x mw) then tf.round (model.py:558-589)
    const int msz = mh * mw;
    float* tm = tmasks + (long long)b * R * msz;
    const int per = (R * msz + (int)gridDim.y - 1) / (int)gridDim.y;
    const int i_lo = (int)blockIdx.y * per, i_hi = i_lo + per < R * msz ? i_lo + per : R * msz;
    for (int i = i_lo + tid; i < i_hi; i += blockDim.x) {
        const int d = i / msz;
        float v = 0.0f;
        if (d < npos) {
            const int r = src_of_dest[d];
            const int g = keep[arg[r]];
            const int py = (i - d * msz) / mw, px = (i - d * msz) % mw;
            const float bx1 = prop[r * 4 + 0], by1 = prop[r * 4 + 1], bx2 = prop[r * 4 + 2], by2 = prop[r * 4 + 3];
            float iny, inx;
            const float sy = (by2 - by1) * (float)(H - 1) / (float)(mh - 1);
            iny = by1 * (float)(H - 1) + (float)py * sy;
            const float sx = (bx2 - bx1) * (float)(W - 1) / (float)(mw - 1);
            inx = bx1 * (float)(W - 1) + (float)px * sx;
            const bool ok = !(iny < 0.0f || iny > (float)(H - 1)) && !(inx < 0.0f || inx > (float)(W - 1));
            if (ok) {
                const int ty = (int)floorf(iny), byy = (int)ceilf(iny);
                const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
                const float wy = iny - (float)ty, wx = inx - (float)lx;
                const uint8_t* mp = gt_masks + (long long)b * H * W * T + g;
                const float tl = (float)(mp[((long long)ty * W + lx) * T] != 0), tr = (float)(mp[((long long)ty * W + rx) * T] != 0);
                const float bl = (float)(mp[((long long)byy * W + lx) * T] != 0), br = (float)(mp[((long long)byy * W + rx) * T] != 0);
                const float top = tl + (tr - tl) * wx;
                const float bot = bl + (br - bl) * wx;
                v = rintf(top + (bot - top) * wy);          // round half to even
            }
        }
        tm[i] = v;
    }
}

// ---------------------------------------------------------------------------------------
// YOLO loss, forward + gradient.  Single block (B*G*G*A is a few thousand boxes).
// ---------------------------------------------------------------------------------------
struct IouOut { float iou, inter, uni, dw, dh; float pminx, pminy, pmaxx, pmaxy, tminx, tminy, tmaxx, tmaxy; };

__device__ __forceinline__ float iou_centre(float px, float py, float pw, float ph, float tx, float ty, float tw, float th, IouOut* o)
{
    const float pminx = px - pw / 2.0f, pmaxx = px + pw / 2.0f, pminy = py - ph / 2.0f, pmaxy = py + ph / 2.0f;
    const float tminx = tx - tw / 2.0f, tmaxx = tx + tw / 2.0f, tminy = ty - th / 2.0f, tmaxy = ty + th / 2.0f;
    const float dw = fminf(pmaxx, tmaxx) - fmaxf(pminx, tminx);
    const float dh = fminf(pmaxy, tmaxy) - fmaxf(pminy, tminy);
    const float iw = fmaxf(dw, 0.0f), ih = fmaxf(dh, 0.0f);
    const float inter = iw * ih;
    const float uni = pw * ph + tw * th - inter;
    const float iou = inter / uni;
    if (o) {
        o->iou = iou; o->inter = inter; o->uni = uni; o->dw = dw; o->dh = dh;
        o->pminx = pminx; o->pminy = pminy; o->pmaxx = pmaxx; o->pmaxy = pmaxy;
        o->tminx = tminx; o->tminy = tminy; o->tmaxx = tmaxx; o->tmaxy = tmaxy;
    }
    return iou;
}

struct LossArgs {
    const float* yt; const float* yp; const float* tb; const float* anchors; const float* cw;
    float obj, noobj, coord, cls, lw;
    float* out; float* grad;
    int B, G, A, C, T;
    int warm;                  // the warm-up branch of model.py:193-207
};

#define LOSS_NACC 9   // sum_xy, sum_wh, sum_conf, sum_cls, n_coord, n_conf, n_cls, nb_true, nb_pred
__global__ __launch_bounds__(256) void yolo_loss_kernel(LossArgs a)
{
    __shared__ double red[LOSS_NACC][256];
    __shared__ double tot[LOSS_NACC];
    const int D = 5 + a.C;
    const int total = a.B * a.G * a.G * a.A;
    const int tid = threadIdx.x;
    double acc[LOSS_NACC];
    for (int k = 0; k < LOSS_NACC; ++k) acc[k] = 0;

    for (int pass = 0; pass < 2; ++pass) {
        for (int i = tid; i < total; i += blockDim.x) {
            const int an = i % a.A;
            const int col = (i / a.A) % a.G;
            const int row = (i / (a.A * a.G)) % a.G;
            const int b = i / (a.A * a.G * a.G);
            const float* p = a.yp + (long long)i * D;
            const float* t = a.yt + (long long)i * D;
            const float sx = myolo_sigmoidf(p[0]), sy = myolo_sigmoidf(p[1]);
            const float px = sx + (float)col, py = sy + (float)row;
            const float pw = myolo_expf(p[2]) * a.anchors[2 * an], ph = myolo_expf(p[3]) * a.anchors[2 * an + 1];
            const float pc = myolo_sigmoidf(p[4]);
            const float t4 = t[4];
            IouOut io;
            const float iou1 = iou_centre(px, py, pw, ph, t[0], t[1], t[2], t[3], &io);
            const float tconf = iou1 * t4;
            int tcls = 0;
            float tv = t[5];
            for (int k = 1; k < a.C; ++k)
                if (t[5 + k] > tv) { tv = t[5 + k]; tcls = k; }
            float best = -INFINITY;
            for (int k = 0; k < a.T; ++k) {
                const float* q = a.tb + ((long long)b * a.T + k) * 4;
                const float v = iou_centre(px, py, pw, ph, q[0], q[1], q[2], q[3], nullptr);
                if (v > best) best = v;
            }
            float coord_mask = t4 * a.coord;
            // warm-up (model.py:193-207, while seen < WARM_UP_BATCHES): a predictor without a box is pulled to its cell centre and its anchor's size,
            // and every predictor's coordinate terms count with weight 1; the IoU / confidence / class terms above keep the plain targets
            float txl = t[0], tyl = t[1], twl = t[2], thl = t[3];
            if (a.warm) {
                const float nob = (coord_mask < a.coord / 2.0f) ? 1.0f : 0.0f;
                txl = t[0] + (0.5f + (float)col) * nob;
                tyl = t[1] + (0.5f + (float)row) * nob;
                twl = t[2] + 1.0f * a.anchors[2 * an] * nob;
                thl = t[3] + 1.0f * a.anchors[2 * an + 1] * nob;
                coord_mask = 1.0f;
            }
            const float conf_mask = ((best < 0.6f) ? 1.0f : 0.0f) * (1.0f - t4) * a.noobj + t4 * a.obj;
            const float class_mask = t4 * a.cw[tcls] * a.cls;
            // softmax CE over class logits
            float mx = p[5];
            for (int k = 1; k < a.C; ++k) mx = fmaxf(mx, p[5 + k]);
            float se = 0.0f;
            for (int k = 0; k < a.C; ++k) se += expf(p[5 + k] - mx);
            const float lse = logf(se) + mx;
            if (pass == 0) {
                const float ex = txl - px, ey = tyl - py, ew = twl - pw, eh = thl - ph, ec = tconf - pc;
                acc[0] += (double)((ex * ex + ey * ey) * coord_mask);
                acc[1] += (double)((ew * ew + eh * eh) * coord_mask);
                acc[2] += (double)(ec * ec * conf_mask);
                acc[3] += (double)((lse - p[5 + tcls]) * class_mask);
                acc[4] += coord_mask > 0.0f;
                acc[5] += conf_mask > 0.0f;
                acc[6] += class_mask > 0.0f;
                acc[7] += (double)t4;
                acc[8] += (tconf > 0.5f && pc > 0.3f) ? 1.0 : 0.0;
            } else {
                const float ncoord = (float)tot[4] + 1e-6f, nconf = (float)tot[5] + 1e-6f, ncls = (float)tot[6] + 1e-6f;
                float* g = a.grad + (long long)i * D;
                float dpx = -(txl - px) * coord_mask / ncoord, dpy = -(tyl - py) * coord_mask / ncoord;
                float dpw = -(twl - pw) * coord_mask / ncoord, dph = -(thl - ph) * coord_mask / ncoord;
                const float dtconf = (tconf - pc) * conf_mask / nconf;
                const float dpc = -dtconf;
                const float diou = dtconf * t4;
                if (diou != 0.0f) {
                    const float U = io.uni, I = io.inter;
                    const float dI = diou / U + diou * I / (U * U);
                    const float dUp = -diou * I / (U * U);
                    const float iw = fmaxf(io.dw, 0.0f), ih = fmaxf(io.dh, 0.0f);
                    const float ddw = (io.dw >= 0.0f) ? dI * ih : 0.0f;      // tf.maximum(d, 0.): first arg on ties
                    const float ddh = (io.dh >= 0.0f) ? dI * iw : 0.0f;
                    const float dpmaxx = (io.pmaxx <= io.tmaxx) ? ddw : 0.0f, dpmaxy = (io.pmaxy <= io.tmaxy) ? ddh : 0.0f;
                    const float dpminx = (io.pminx >= io.tminx) ? -ddw : 0.0f, dpminy = (io.pminy >= io.tminy) ? -ddh : 0.0f;
                    dpx += dpmaxx + dpminx;
                    dpy += dpmaxy + dpminy;
                    dpw += (dpmaxx - dpminx) / 2.0f + dUp * ph;
                    dph += (dpmaxy - dpminy) / 2.0f + dUp * pw;
                }
                g[0] = dpx * sx * (1.0f - sx) * a.lw;
                g[1] = dpy * sy * (1.0f - sy) * a.lw;
                g[2] = dpw * pw * a.lw;
                g[3] = dph * ph * a.lw;
                g[4] = dpc * pc * (1.0f - pc) * a.lw;
                const float cm = class_mask / ncls;
                for (int k = 0; k < a.C; ++k) {
                    const float sm = expf(p[5 + k] - mx) / se;
                    g[5 + k] = (sm - (k == tcls ? 1.0f : 0.0f)) * cm * a.lw;
                }
            }
        }
        if (pass == 0) {
            for (int k = 0; k < LOSS_NACC; ++k) red[k][tid] = acc[k];
            __syncthreads();
            for (int o = blockDim.x / 2; o > 0; o >>= 1) {
                if (tid < o)
                    for (int k = 0; k < LOSS_NACC; ++k) red[k][tid] += red[k][tid + o];
                __syncthreads();
            }
            if (tid < LOSS_NACC) tot[tid] = red[tid][0];
            __syncthreads();
            if (tid == 0) {
                const float ncoord = (float)tot[4], nconf = (float)tot[5], ncls = (float)tot[6];
                const float lxy = (float)tot[0] / (ncoord + 1e-6f) / 2.0f;
                const float lwh = (float)tot[1] / (ncoord + 1e-6f) / 2.0f;
                const float lcf = (float)tot[2] / (nconf + 1e-6f) / 2.0f;
                const float lcl = (float)tot[3] / (ncls + 1e-6f);
                a.out[0] = lxy + lwh + lcf + lcl;
                a.out[1] = lxy; a.out[2] = lwh; a.out[3] = lcf; a.out[4] = lcl;
                a.out[5] = (float)tot[8] / ((float)tot[7] + 1e-6f);
                a.out[6] = ncoord; a.out[7] = nconf;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// unmold_mask for every detection of one image (myolo_utils.py:883-912 inside MaskYOLO.decode_masks,
// model.py:1355-1389): pick the class channel, resize the mh x mw mask to the detection's clamped pixel
// window (order-1, pixel centres aligned, edge clamp -- float32, this operation order), threshold at 0.5,
// paste.  out[H][W][N] uint8 (the reference's np.stack(axis=-1) layout); lanes run along N so a wave writes
// N contiguous bytes per pixel.
// ---------------------------------------------------------------------------------------
// skimage.transform.resize(..., clip=True) clips its output to the value range of the INPUT mask.  With the zero border of
// mode='constant' that matters in exactly one case for the 0.5 threshold: a mask whose minimum is already >= 0.5 keeps its rim
// (the fade towards 0 is clipped back up to the minimum).  allhigh[n] = 1 for such a detection (its class channel).
__global__ __launch_bounds__(256) void unmold_allhigh_kernel(const float* __restrict__ masks, const float* __restrict__ det,
                                                             int32_t* __restrict__ allhigh, int mh, int mw, int C)
{
    __shared__ int low;
    const int n = blockIdx.x;
    if (threadIdx.x == 0) low = 0;
    __syncthreads();
    const int cls = (int)det[(long long)n * 6 + 5];
    const float* m = masks + (long long)n * mh * mw * C + cls;
    int mine = 0;
    for (int i = threadIdx.x; i < mh * mw; i += blockDim.x)
        if (!(m[(long long)i * C] >= 0.5f)) mine = 1;
    if (mine) low = 1;
    __syncthreads();
    if (threadIdx.x == 0) allhigh[n] = low ? 0 : 1;
}

__global__ __launch_bounds__(256) void unmold_kernel(const float* __restrict__ masks, const float* __restrict__ det,
                                                     const int32_t* __restrict__ allhigh, uint8_t* __restrict__ out, int N, int mh, int mw,
                                                     int C, int H, int W)
{
    const long long total = (long long)H * W * N;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int n = (int)(i % N);
        const int pix = (int)(i / N);
        const int y = pix / W, x = pix - y * W;
        const float* d = det + (long long)n * 6;
        // image_shape[0] is used for x and [1] for y (myolo_utils.py:892); square images here
        int x1 = min(max(0, (int)(d[0] * (float)W)), W), x2 = min(max(1, (int)(d[2] * (float)W)), W);
        int y1 = min(max(0, (int)(d[1] * (float)H)), H), y2 = min(max(1, (int)(d[3] * (float)H)), H);
        uint8_t v = 0;
        if (y >= y1 && y < y2 && x >= x1 && x < x2) {
            const int oh = max(1, y2 - y1), ow = max(1, x2 - x1);
            const int cls = (int)d[5];
            const float sy = (float)mh / (float)oh, sx = (float)mw / (float)ow;
            float fy = ((float)(y - y1) + 0.5f) * sy - 0.5f;
            float fx = ((float)(x - x1) + 0.5f) * sx - 0.5f;
            // skimage.transform.resize(order=1, mode='constant', cval=0) as the reference calls it (myolo_utils.py:433-447, 903):
            // samples outside the 28x28 mask read 0 (NOT the edge value), so an up-scaled mask fades to 0 over the outermost
            // half source pixel -- pinned against scikit-image 0.18.3 in tests/golden/skimage_resize_fixture.npz
            const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            const int yb = y0 + 1, xb = x0 + 1;
            const float wy = fy - (float)y0, wx = fx - (float)x0;
            const float* m = masks + (long long)n * mh * mw * C + cls;
            const bool y0in = (unsigned)y0 < (unsigned)mh, ybin = (unsigned)yb < (unsigned)mh;
            const bool x0in = (unsigned)x0 < (unsigned)mw, xbin = (unsigned)xb < (unsigned)mw;
            const float tl = (y0in && x0in) ? m[((long long)y0 * mw + x0) * C] : 0.f, tr = (y0in && xbin) ? m[((long long)y0 * mw + xb) * C] : 0.f;
            const float bl = (ybin && x0in) ? m[((long long)yb * mw + x0) * C] : 0.f, br = (ybin && xbin) ? m[((long long)yb * mw + xb) * C] : 0.f;
            const float top = tl + (tr - tl) * wx;
            const float bot = bl + (br - bl) * wx;
            v = ((top + (bot - top) * wy) >= 0.5f || allhigh[n]) ? 1 : 0;        // clip=True: see unmold_allhigh_kernel
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------
// GPU producer of the Shapes input pipeline: rasterise the images and instance masks
// of a batch from their shape specifications (example/shapes/dataset_shapes.py:80-135: load_image / load_mask /
// draw_shape incl. the occlusion rule :112-116), drop empty instances and take tight boxes (load_image_gt + extract_bboxes,
// myolo_utils.py:247-271,346-352), and encode the YOLO targets (BatchGenerator.__getitem__, myolo_utils.py:753-844).
// spec layout per image (int32): [bg_r, bg_g, bg_b, n_shapes] then SHAPE_INTS per shape:
//   [type(1 square, 2 circle, 3 triangle), r, g, b, x, y, s, ax, ay, bx, by, cx, cy]  (triangle vertices already int-truncated)
// ---------------------------------------------------------------------------------------
#define SHAPE_INTS 13
#define SPEC_MAXS 8

__device__ __forceinline__ bool shape_covers(const int* sp, int px, int py)
{
    const int type = sp[0], x = sp[4], y = sp[5], s = sp[6];
    if (type == 1) return px >= x - s && px <= x + s && py >= y - s && py <= y + s;
    if (type == 2) return (px - x) * (px - x) + (py - y) * (py - y) <= s * s;
    const int ax = sp[7], ay = sp[8], bx = sp[9], by = sp[10], cx = sp[11], cy = sp[12];
    const int e0 = (px - ax) * (by - ay) - (py - ay) * (bx - ax);
    const int e1 = (px - bx) * (cy - by) - (py - by) * (cx - bx);
    const int e2 = (px - cx) * (ay - cy) - (py - cy) * (ax - cx);
    return (e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0);
}

__global__ void shapes_stats_init_kernel(int* stats, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { stats[i * 5 + 0] = 0; stats[i * 5 + 1] = 0x7fffffff; stats[i * 5 + 2] = -1; stats[i * 5 + 3] = 0x7fffffff; stats[i * 5 + 4] = -1; }
}

// pass 1: per (image, shape) visible-pixel count and extent.  stats[b][s] = {count, minx, maxx, miny, maxy}
__global__ __launch_bounds__(256) void shapes_stats_kernel(const int* __restrict__ spec, int spec_stride, int* __restrict__ stats,
                                                           int H, int W, int S)
{
    const int b = blockIdx.y;
    const int* sp = spec + (long long)b * spec_stride;
    const int n = sp[3];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= H * W) return;
    const int py = pix / W, px = pix - py * W;
    bool later = false;
    for (int k = n - 1; k >= 0; --k) {          // walk from the last drawn shape: it occludes the earlier ones
        const bool in = shape_covers(sp + 4 + k * SHAPE_INTS, px, py);
        if (in && !later) {
            int* st = stats + ((long long)b * S + k) * 5;
            atomicAdd(st + 0, 1);
            atomicMin(st + 1, px); atomicMax(st + 2, px);
            atomicMin(st + 3, py); atomicMax(st + 4, py);
        }
        later = later || in;
    }
}

// pass 2 (one workgroup per image): compaction of non-empty instances, boxes, class ids, YOLO target encoding
__global__ void shapes_encode_kernel(const int* __restrict__ spec, int spec_stride, const int* __restrict__ stats, int* __restrict__ slotmap,
                                     int32_t* __restrict__ gt_ids, int32_t* __restrict__ gt_boxes, float* __restrict__ y_true,
                                     float* __restrict__ true_boxes, const double* __restrict__ anchors, int H, int W, int S, int T,
                                     int G, int A, int C)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int* sp = spec + (long long)b * spec_stride;
    const int n = sp[3];
    const double cell_w = (double)W / G, cell_h = (double)H / G;
    int m = 0, tbi = 0;
    for (int k = 0; k < S; ++k) slotmap[b * S + k] = -1;
    for (int k = 0; k < n; ++k) {
        const int* st = stats + ((long long)b * S + k) * 5;
        if (st[0] <= 0 || m >= T) continue;                   // np.sum(mask) > 0 filter (myolo_utils.py:346)
        slotmap[b * S + k] = m;
        const int x1 = st[1], x2 = st[2] + 1, y1 = st[3], y2 = st[4] + 1;     // extract_bboxes: x2, y2 exclusive
        const int cls = sp[4 + k * SHAPE_INTS];
        gt_ids[b * T + m] = cls;
        int32_t* gb = gt_boxes + ((long long)b * T + m) * 4;
        gb[0] = x1; gb[1] = y1; gb[2] = x2; gb[3] = y2;
        const double cx = .5 * (x1 + x2) / cell_w, cy = .5 * (y1 + y2) / cell_h;
        const int gx = (int)floor(cx), gy = (int)floor(cy);
        if (gx < G && gy < G) {
            const double bw = (x2 - x1) / cell_w, bh = (y2 - y1) / cell_h;
            int best = -1;
            double max_iou = -1;
            for (int j = 0; j < A; ++j) {
                const double aw = anchors[2 * j], ah = anchors[2 * j + 1];
                const double iw = fmin(bw, aw), ih = fmin(bh, ah);
                const double inter = iw * ih;
                const double iou = inter / (bw * bh + aw * ah - inter);
                if (max_iou < iou) { best = j; max_iou = iou; }
            }
            float* yt = y_true + ((((long long)b * G + gy) * G + gx) * A + best) * (5 + C);
            yt[0] = (float)cx; yt[1] = (float)cy; yt[2] = (float)bw; yt[3] = (float)bh; yt[4] = 1.0f;
            yt[5 + cls] = 1.0f;
            float* tb = true_boxes + ((long long)b * T + tbi) * 4;
            tb[0] = (float)cx; tb[1] = (float)cy; tb[2] = (float)bw; tb[3] = (float)bh;
            tbi = (tbi + 1) % T;
        }
        ++m;
    }
}

// pass 3: pixels -- image (uint8 colour / 255 through a host-provided table) and instance masks in compacted order
__global__ __launch_bounds__(256) void shapes_render_kernel(const int* __restrict__ spec, int spec_stride, const int* __restrict__ slotmap,
                                                            const float* __restrict__ lut, float* __restrict__ images,
                                                            uint8_t* __restrict__ masks, int H, int W, int S, int T)
{
    const int b = blockIdx.y;
    const int* sp = spec + (long long)b * spec_stride;
    const int n = sp[3];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= H * W) return;
    const int py = pix / W, px = pix - py * W;
    int r = sp[0], g = sp[1], bl = sp[2];
    uint8_t* mp = masks + ((long long)b * H * W + pix) * T;
    for (int t = 0; t < T; ++t) mp[t] = 0;
    bool later = false;
    for (int k = n - 1; k >= 0; --k) {
        const int* s1 = sp + 4 + k * SHAPE_INTS;
        const bool in = shape_covers(s1, px, py);
        if (in && !later) {
            r = s1[1]; g = s1[2]; bl = s1[3];               // the top-most shape gives the pixel its colour
            const int slot = slotmap[b * S + k];
            if (slot >= 0) mp[slot] = 1;
        }
        later = later || in;
    }
    float* ip = images + ((long long)b * H * W + pix) * 3;
    ip[0] = lut[r & 255]; ip[1] = lut[g & 255]; ip[2] = lut[bl & 255];
}

extern "C" {

int myolo_yolo_decode(const float* y_pred, const float* anchors, float* proposals, int B, int G, int A, int C, void* stream)
{
    MYOLO_REQUIRE(y_pred && anchors && proposals && B > 0 && G > 0 && A > 0 && C > 0, "yolo_decode: bad arguments");
    const int total = B * G * G * A;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, y_pred, anchors,
                       proposals, total, G, A, 5 + C, 0);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_yolo_detections(const float* y_pred, const float* anchors, float* detections, int B, int G, int A, int C, void* stream)
{
    MYOLO_REQUIRE(y_pred && anchors && detections && B > 0 && G > 0 && A > 0 && C > 0, "yolo_detections: bad arguments");
    const int total = B * G * G * A;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, y_pred, anchors,
                       detections, total, G, A, 5 + C, 1);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_shapes_batch(const int32_t* spec, int spec_stride, const double* anchors, const float* lut, float* images, uint8_t* gt_masks,
                       int32_t* gt_boxes, int32_t* gt_class_ids, float* y_true, float* true_boxes, int B, int H, int W, int S, int T,
                       int G, int A, int C, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(spec && anchors && lut && images && gt_masks && gt_boxes && gt_class_ids && y_true && true_boxes, "shapes_batch: null pointer");
    MYOLO_REQUIRE(B > 0 && H > 0 && W > 0 && S > 0 && S <= SPEC_MAXS && T > 0 && spec_stride >= 4 + S * SHAPE_INTS, "shapes_batch: bad sizes");
    const size_t need = (size_t)B * S * 6 * sizeof(int);
    MYOLO_NEED_WS(need);
    int* stats = (int*)ws;
    int* slotmap = stats + (size_t)B * S * 5;
    hipStream_t s = (hipStream_t)stream;
    // stats init: count 0, min = INT_MAX, max = -1  (pattern fill through a tiny kernel-free trick: two memsets)
    (void)hipMemsetAsync(stats, 0, (size_t)B * S * 5 * sizeof(int), s);
    (void)hipMemsetAsync(gt_boxes, 0, (size_t)B * T * 4 * sizeof(int32_t), s);
    (void)hipMemsetAsync(gt_class_ids, 0, (size_t)B * T * sizeof(int32_t), s);
    (void)hipMemsetAsync(y_true, 0, (size_t)B * G * G * A * (5 + C) * sizeof(float), s);
    (void)hipMemsetAsync(true_boxes, 0, (size_t)B * T * 4 * sizeof(float), s);
    hipLaunchKernelGGL(shapes_stats_init_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, stats, B * S);
    dim3 grid((H * W + 255) / 256, B);
    hipLaunchKernelGGL(shapes_stats_kernel, grid, dim3(256), 0, s, spec, spec_stride, stats, H, W, S);
    hipLaunchKernelGGL(shapes_encode_kernel, dim3(B), dim3(64), 0, s, spec, spec_stride, stats, slotmap, gt_class_ids, gt_boxes, y_true,
                       true_boxes, anchors, H, W, S, T, G, A, C);
    hipLaunchKernelGGL(shapes_render_kernel, grid, dim3(256), 0, s, spec, spec_stride, slotmap, lut, images, gt_masks, H, W, S, T);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_unmold_masks(const float* masks, const float* detections, uint8_t* full_masks, int N, int mh, int mw, int C, int H,
                       int W, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(masks && detections && full_masks && N > 0 && mh > 0 && mw > 0 && C > 0 && H > 0 && W > 0, "unmold_masks: bad arguments");
    MYOLO_NEED_WS((size_t)N * sizeof(int32_t));
    int32_t* allhigh = (int32_t*)ws;
    hipLaunchKernelGGL(unmold_allhigh_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, masks, detections, allhigh, mh, mw, C);
    const long long total = (long long)H * W * N;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unmold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, masks, detections, allhigh, full_masks, N,
                       mh, mw, C, H, W);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_mask_targets(const float* proposals, const int32_t* gt_class_ids, const int32_t* gt_boxes_px, const uint8_t* gt_masks,
                       float* rois, int32_t* target_class_ids, float* target_masks, int32_t* n_pos, int B, int R, int T, int H,
                       int W, int mh, int mw, void* stream)
{
    MYOLO_REQUIRE(proposals && gt_class_ids && gt_boxes_px && gt_masks && rois && target_class_ids && target_masks && n_pos,
                  "mask_targets: null pointer");
    MYOLO_REQUIRE(B > 0 && R > 0 && R <= MT_MAXR && T > 0 && T <= MT_MAXT && mh > 1 && mw > 1,
                  "mask_targets: need R<=%d, T<=%d, mask shape > 1", MT_MAXR, MT_MAXT);
    hipLaunchKernelGGL(mask_targets_kernel, dim3(B, B >= 128 ? 2 : 8), dim3(256), 0, (hipStream_t)stream, proposals, gt_class_ids, gt_boxes_px,
                       gt_masks, rois, target_class_ids, target_masks, n_pos, R, T, H, W, mh, mw);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_yolo_loss(const float* y_true, const float* y_pred, const float* true_boxes, const float* anchors,
                    const float* class_weights, float object_scale, float no_object_scale, float coord_scale, float class_scale,
                    float loss_weight, float* out_terms, float* grad, int B, int G, int A, int C, int T, void* ws, size_t ws_bytes,
                    void* stream)
{
    return myolo_yolo_loss_warmup(y_true, y_pred, true_boxes, anchors, class_weights, object_scale, no_object_scale, coord_scale, class_scale,
                                  loss_weight, 0, out_terms, grad, B, G, A, C, T, ws, ws_bytes, stream);
}

int myolo_yolo_loss_warmup(const float* y_true, const float* y_pred, const float* true_boxes, const float* anchors,
                           const float* class_weights, float object_scale, float no_object_scale, float coord_scale, float class_scale,
                           float loss_weight, int warmup, float* out_terms, float* grad, int B, int G, int A, int C, int T, void* ws, size_t ws_bytes,
                           void* stream)
{
    (void)ws; (void)ws_bytes;
    MYOLO_REQUIRE(y_true && y_pred && true_boxes && anchors && class_weights && out_terms && grad, "yolo_loss: null pointer");
    MYOLO_REQUIRE(B > 0 && G > 0 && A > 0 && C > 0 && T > 0, "yolo_loss: bad sizes");
    LossArgs a{y_true, y_pred, true_boxes, anchors, class_weights, object_scale, no_object_scale, coord_scale, class_scale,
               loss_weight, out_terms, grad, B, G, A, C, T, warmup ? 1 : 0};
    hipLaunchKernelGGL(yolo_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"
