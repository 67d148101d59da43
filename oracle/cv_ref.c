/* cv_ref.c — CPU restatement of the two OpenCV calls on the reference's crop path.  TEST INFRASTRUCTURE ONLY.
 *
 * Reference call sites (model_training/utils/utils.py):
 *   :246-248  cv2.copyMakeBorder(crop, top, bottom, left, right, cv2.BORDER_CONSTANT, value=padding_value)
 *   :235,251  albumentations.Resize(crop_size, crop_size)  ==  cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)
 * Pinned third-party versions (requirements.txt:1,11): opencv-python 4.4.0.42, albumentations 1.0.0.  Neither package
 * exists in this image or on the GPU box, so this file restates the PUBLISHED algorithm of OpenCV 4.4's
 * modules/imgproc/src/resize.cpp for CV_8U + INTER_LINEAR (the generic fixed-point path: IPP declines 8u linear unless
 * useIPP_NotExact(), the SIMD row/column functors are bit-identical to the scalar ones) and of copyMakeBorder's
 * constant mode — "parity unpinned" against OpenCV itself, said here and in DESIGN.md.
 *
 * It is deliberately written the way resize.cpp is structured — coefficient TABLES built once per call
 * (xofs / ialpha / yofs / ibeta as short), a horizontal pass over source rows into int rows (HResizeLinear), a vertical
 * pass with FixedPtCast-style arithmetic (VResizeLinear for 8u) — and shares no code or formulation with the product's
 * feartracker_amd/geometry.py (vectorised numpy, weights as int64 arrays) or with crop_resize_normalize_kernel (one
 * thread per output pixel, taps computed on the fly).  The three are compared bit for bit in tests/.
 *
 * Arithmetic rules restated (resize.cpp, OpenCV 4.4):
 *   scale_x = 1 / ((double)dw / sw)                      (inv_scale_x = dsize.width / ssize.width, scale = 1 / inv_scale)
 *   fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = cvFloor(fx);  fx -= sx;          (float subtraction)
 *   sx < 0          -> fx = 0, sx = 0
 *   sx + 1 >= sw    -> xmax = min(xmax, dx);  if (sx >= sw - 1) fx = 0, sx = sw - 1
 *   ialpha[2dx]   = saturate_cast<short>((1.f - fx) * 2048),  ialpha[2dx+1] = saturate_cast<short>(fx * 2048)
 *                   (INTER_RESIZE_COEF_BITS = 11; saturate_cast<short>(float) = cvRound = round half to even)
 *   rows: fy, sy, ibeta likewise but NOT clamped in the table; the two source rows are clip(sy, 0, sh), clip(sy + 1, 0, sh)
 *         with clip(x, a, b) = x >= a ? (x < b ? x : b - 1) : a, the weights stay what the table says
 *   HResizeLinear:  D[dx] = S[sx] * a0 + S[sx + cn] * a1   for dx < xmax,   D[dx] = S[sx] * 2048   for dx >= xmax
 *   VResizeLinear (8u):  dst = uchar(( ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
 *   cv::resize special cases: dsize == ssize -> copy;  INTER_LINEAR with an exact 2x2 decimation is executed as the
 *   fast INTER_AREA (2x2 box mean, (a + b + c + d + 2) >> 2)
 *   copyMakeBorder constant: every channel of the border = saturate_cast<uchar>(double) = clamp(cvRound(v), 0, 255)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define COEF_BITS 11
#define COEF_SCALE (1 << COEF_BITS)

static int cv_floor_f(float v) { return (int)floorf(v); }

/* cvRound: nearest, ties to even (the default FP rounding mode of lrint) */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

static short sat_short(float v) {
    int r = cv_round_f(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

static int clip_row(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

/* horizontal pass of one source row: S [sw*cn] uchar -> D [dw*cn] int */
static void hresize_linear_row(const uint8_t* S, int* D, const int* xofs, const short* alpha, int dwidth_cn, int xmax_cn, int cn) {
    int dx = 0;
    for (; dx < xmax_cn; ++dx) {
        const int sx = xofs[dx];
        D[dx] = S[sx] * alpha[dx * 2] + S[sx + cn] * alpha[dx * 2 + 1];
    }
    for (; dx < dwidth_cn; ++dx) D[dx] = S[xofs[dx]] * COEF_SCALE;
}

static void vresize_linear_8u(const int* S0, const int* S1, uint8_t* dst, short b0, short b1, int width_cn) {
    for (int x = 0; x < width_cn; ++x)
        dst[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
}

/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) for CV_8UC(cn); sstep / dstep in bytes.  Returns 0, or -1 on bad
 * arguments / allocation failure. */
int cvref_resize_linear_8u(const uint8_t* src, int sh, int sw, int cn, long sstep, uint8_t* dst, int dh, int dw, long dstep) {
    if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0 || cn > 4) return -1;
    if (sh == dh && sw == dw) {
        for (int y = 0; y < sh; ++y) memcpy(dst + (long)y * dstep, src + (long)y * sstep, (size_t)sw * cn);
        return 0;
    }
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    {   /* exact 2x2 decimation: INTER_LINEAR is replaced by the fast INTER_AREA */
        const int iscale_x = (int)lrint(scale_x) > 0 ? (int)lrint(scale_x) : 1;      /* saturate_cast<int>(scale_x) */
        const int iscale_y = (int)lrint(scale_y) > 0 ? (int)lrint(scale_y) : 1;
        const int fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
        if (fast && iscale_x == 2 && iscale_y == 2) {
            for (int y = 0; y < dh; ++y) {
                const uint8_t* r0 = src + (long)(2 * y) * sstep;
                const uint8_t* r1 = src + (long)(2 * y + 1) * sstep;
                uint8_t* d = dst + (long)y * dstep;
                for (int x = 0; x < dw; ++x)
                    for (int c = 0; c < cn; ++c)
                        d[x * cn + c] = (uint8_t)((r0[2 * x * cn + c] + r0[(2 * x + 1) * cn + c] + r1[2 * x * cn + c] + r1[(2 * x + 1) * cn + c] + 2) >> 2);
            }
            return 0;
        }
    }
    const int dwc = dw * cn;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dwc);
    short* ialpha = (short*)malloc(sizeof(short) * (size_t)dwc * 2);
    int* yofs = (int*)malloc(sizeof(int) * (size_t)dh);
    short* ibeta = (short*)malloc(sizeof(short) * (size_t)dh * 2);
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * (size_t)dwc);
    rows[1] = (int*)malloc(sizeof(int) * (size_t)dwc);
    if (!xofs || !ialpha || !yofs || !ibeta || !rows[0] || !rows[1]) {
        free(xofs); free(ialpha); free(yofs); free(ibeta); free(rows[0]); free(rows[1]);
        return -1;
    }
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        for (int k = 0; k < cn; ++k) xofs[dx * cn + k] = sx * cn + k;
        const float cbuf[2] = {1.f - fx, fx};
        for (int k = 0; k < 2; ++k) ialpha[dx * cn * 2 + k] = sat_short(cbuf[k] * COEF_SCALE);
        for (int k = 2; k < cn * 2; ++k) ialpha[dx * cn * 2 + k] = ialpha[dx * cn * 2 + k - 2];
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short((1.f - fy) * COEF_SCALE);
        ibeta[dy * 2 + 1] = sat_short(fy * COEF_SCALE);
    }
    /* row loop: the two int rows are a cache keyed by source row (resizeGeneric_Invoker keeps the rows it already
     * converted for the previous destination row) */
    int tag[2] = {-1, -1};
    for (int dy = 0; dy < dh; ++dy) {
        const int sy0 = yofs[dy];
        const int* use[2];
        int pinned = -1;                                    /* buffer holding the first wanted row */
        for (int k = 0; k < 2; ++k) {
            const int want = clip_row(sy0 + k, 0, sh);
            int j = tag[0] == want ? 0 : (tag[1] == want ? 1 : -1);
            if (j < 0) {
                j = pinned == 0 ? 1 : 0;
                hresize_linear_row(src + (long)want * sstep, rows[j], xofs, ialpha, dwc, xmax * cn, cn);
                tag[j] = want;
            }
            if (k == 0) pinned = j;
            use[k] = rows[j];
        }
        vresize_linear_8u(use[0], use[1], dst + (long)dy * dstep, ibeta[dy * 2], ibeta[dy * 2 + 1], dwc);
    }
    free(xofs); free(ialpha); free(yofs); free(ibeta); free(rows[0]); free(rows[1]);
    return 0;
}

/* cv::copyMakeBorder(src, dst, top, bottom, left, right, BORDER_CONSTANT, Scalar(value[0..3])) for CV_8UC(cn).
 * dst must hold (sh + top + bottom) x (sw + left + right) x cn bytes, dstep bytes per row. */
int cvref_copy_make_border_const_8u(const uint8_t* src, int sh, int sw, int cn, long sstep, uint8_t* dst, long dstep,
                                    int top, int bottom, int left, int right, const double* value) {
    if (!dst || sh < 0 || sw < 0 || cn <= 0 || cn > 4 || top < 0 || bottom < 0 || left < 0 || right < 0 || !value) return -1;
    uint8_t fill[4];
    for (int c = 0; c < cn; ++c) {
        const int r = cv_round_d(value[c]);
        fill[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    const int dh = sh + top + bottom, dw = sw + left + right;
    for (int y = 0; y < dh; ++y) {
        uint8_t* d = dst + (long)y * dstep;
        const int inside = y >= top && y < top + sh;
        for (int x = 0; x < dw; ++x) {
            if (inside && x >= left && x < left + sw) {
                x += sw - 1;          /* the interior is copied below in one piece */
                continue;
            }
            for (int c = 0; c < cn; ++c) d[x * cn + c] = fill[c];
        }
        if (inside && sw > 0) memcpy(d + (long)left * cn, src + (long)(y - top) * sstep, (size_t)sw * cn);
    }
    return 0;
}

/* the coefficient tables alone (tests compare them with the product's taps): idx[dst], w0[dst], w1[dst] for one axis,
 * clamp = 1 for the column rule (table clamped), 0 for the row rule (raw sy, weights unclamped) */
int cvref_linear_table(int dst, int src, int clamp, int* idx, int* w0, int* w1) {
    if (dst <= 0 || src <= 0 || !idx || !w0 || !w1) return -1;
    const double scale = 1. / ((double)dst / src);
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor_f(f);
        f -= s;
        if (clamp) {
            if (s < 0) { f = 0; s = 0; }
            if (s + 1 >= src && s >= src - 1) { f = 0; s = src - 1; }
        }
        idx[d] = s;
        w0[d] = sat_short((1.f - f) * COEF_SCALE);
        w1[d] = sat_short(f * COEF_SCALE);
    }
    return 0;
}
