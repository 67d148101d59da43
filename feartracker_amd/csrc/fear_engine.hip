// fear_engine.hip — host side of the MI355X FEAR engine: model-file parsing, weight packing,
// launch-plan construction, workspace management and the C ABI of include/fear_hip.h.
//
// Reference behaviour reproduced (BN folded, inference only):
//   FEARNet.get_features  model_training/model/fear_net.py:63-66  (Encoder.stages[:4] + AdjustLayer)
//   FEARNet.track         model_training/model/fear_net.py:90-96  (+ BoxTower.forward, model/blocks.py:174-194)
//   FEARBoxCoder.decode   model_training/dataset/box_coder.py:75-107
// The launch plan is derived from the block table of the .fearw file (include/fearw_format.h), so any
// FBNet-style trunk (stem + inverted-residual blocks) with the FEAR head runs, not only FEAR-XS.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fear_hip.h"
#include "../../include/fearw_format.h"
#include "fear_kernels.h"
#include "fear_headchain.h"
#include "fear_headchain_b.h"
#include "fear_e1pair.h"
#include "fear_chain32.h"
#ifndef FEAR_E1PAIR_TPW_MAX
#define FEAR_E1PAIR_TPW_MAX 1      // e1pair_kernel: at most this many consecutive tiles per workgroup (more measured no faster: fear_e1pair.h)
#endif

namespace {

using namespace fear;

// ---------------------------------------------------------------------------------------------
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal: normalise
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - e;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}

uint16_t float_to_bf16(float f) {     // round to nearest even, like v_cvt_pk_bf16_f32
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7f800000u) == 0x7f800000u) return (uint16_t)((x >> 16) | ((x & 0xffffu) ? 0x40u : 0));
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

struct Conv {
    int cout, cin_g, groups, k, stride, pad, relu, has_bias;
    std::vector<float> w;  // OIHW fp32
    std::vector<float> b;
    // device copies
    float* d_w = nullptr;  // pointwise: [N][K]; depthwise: [k*k][C]; stem: [27][16]
    float* d_w2 = nullptr; // stem only: [16][32] zero-padded rows for the implicit-GEMM kernel
    float* d_b = nullptr;
    bool is_dw() const { return groups == cout && cin_g == 1 && groups > 1; }
    bool is_pw() const { return groups == 1 && k == 1; }
};

enum OpType { OP_STEM, OP_PW, OP_DW, OP_CORR, OP_PW_SMALL, OP_IR16, OP_IRTILE, OP_CHAIN16, OP_HEADCHAIN, OP_E1PAIR, OP_CHAIN32, OP_CHAIN32_16 };

struct Op {
    OpType type;
    int conv = -1;
    // tensors: buffer id (-1 = external), channel offset inside the row, row stride (floats)
    int in_buf = -1, in_ld = 0, in_off = 0;
    int out_buf = -1, out_ld = 0, out_off = 0;
    int res_buf = -1, res_ld = 0;
    int H = 0, W = 0;         // input spatial size
    int Ho = 0, Wo = 0;       // output spatial size
    int C = 0, N = 0;         // input / output channels
    int relu = 0, act = 0;
    int out_external = 0;     // 1: features NCHW, 2: bbox, 3: cls
    int tmpl_cls = 0;         // OP_CORR / corr_fused: use the classification template
    int pred_fused = 0;       // OP_IR16 (sep16): the prediction SepConv that consumes this layer runs in its epilogue
    float* pred_packed = nullptr;
    int pred_conv_p = -1;
    int small_tiles = 0;      // OP_IRTILE: 1 = kFusedTileSmall (small-batch plan), 2 = kFusedTileTiny (a handful of crops)
    int pw_split = 0;         // OP_PW / OP_CORR: spread the output-channel passes over gridDim.y workgroups (small-batch plan)
    int splitk = 0;           // OP_IR16: > 0 = workgroups per crop (split over expansion chunks) + a reduce launch
    int tiny = 0;             // OP_IR16 (sep16, 16 output channels per workgroup): sep16_tiny_kernel (Fused16::kernel_tiny)
    int nsplit = 0;           // OP_IR16 (sep16): > 0 = 16-channel output slices, one workgroup each (Ir2Args::nsplit_wstride)
    int part_buf = -1;        //          scratch buffer of the partial projections
    int lane = 0;             // 1: bbox branch of the head, may run on the handle's second stream (small batches)
    int corr_fused = 0;       // OP_IR16 (sep16): the pixel-wise correlation runs in this kernel's epilogue
    int conv_e = -1, conv_d = -1, conv_p = -1;  // OP_IR16: expand (or -1) / depthwise / project convs
    int relu_dw = 0;
    int fused_id = -1;        // OP_IR16: index into the fused-kernel table
    float* d_packed = nullptr; // OP_IR16: per-chunk packed weights (Ir2Geom layout), owned by the handle
    int math = 0;              // OP_IR16: 1 = fp16-split matrix-pipe kernel
    int pred_cout = 0;         // OP_IR16 prediction head: real output channels (4 / 1), NCHW external output
    int stem = 0;              // OP_IRTILE: stem conv fused in front (reads the caller's NCHW image)
    int io_bf16 = 0;           // OP_IRTILE: IO_X_BF16 | IO_Y_BF16 | IO_R_BF16 — which of its tensors are stored in bf16 (kTileBf16)
    // OP_CHAIN16: per-block packed weights / projection convs, neck fragments
    float* chain_pk[8] = {nullptr};
    int chain_cp[8] = {0};
    // OP_CHAIN32_16: chain_pk / chain_cp hold the four 32 x 32 blocks, these the seven stride-16 blocks
    float* chain16_pk[8] = {nullptr};
    int chain16_cp[8] = {0};
    float* neck_pk = nullptr;
    int neck_conv = -1;
    // OP_HEADCHAIN (headchain_kernel): per branch (0 = classification, 1 = regression) the pass-major weights of its four SepConvs,
    // the depthwise weights of the first layer and of the correlation chunks, the packed prediction SepConv and its pointwise conv
    float* hc_w[2][4] = {{nullptr}};
    float* hc_wd0[2] = {nullptr};
    float* hc_wdc[2] = {nullptr};
    float* hc_pred[2] = {nullptr};
    float* hc_taps[2][4] = {{nullptr}};   // math 2 (headchain_b_kernel): each layer's own depthwise taps
    int hc_pred_conv[2] = {-1, -1};
    int hc_pred_act[2] = {0, 0};
    char name[64];
    double flops = 0, bytes = 0;  // per crop: algorithmic FLOPs, compulsory bytes (in + out + weights excluded)
    // profiling
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double prof_ms = 0;
    int64_t prof_n = 0;
};

struct Plan {
    int hw = 0;
    bool with_head = false;
    bool small = false;
    std::vector<Op> ops;
    int head_first = -1;             // index of the first head op when the two branches were planned on disjoint buffers
    int n_bufs = 0;
    size_t buf_floats_per_crop = 0;  // every pool buffer has this many floats per crop
};

}  // namespace

struct fear_handle {
    int device = 0;
    std::vector<Conv> convs;
    std::vector<FearwBlock> blocks;
    bool has_head = false;
    int feat_channels = 0;
    int max_batch = 64;
    int profile = 0;
    int profile_op = -1;   // -1: every op, else only this op index of each plan
    int fuse = 1;          // 1: use the fused block kernels where an instantiation exists
    int tiny_sep = 1;      // FEAR_OPT_TINY_SEP: 1 = the tiny plan's 16-channel SepConv slices run sep16_tiny_kernel
    int tile_v4 = 1;       // FEAR_OPT_TILE_V4: 1 = phase-overlapped tile kernel for the blocks of kFusedTileV4 (throughput plan)
    int chain = 1;         // 1: run the stride-16 trunk stage as one register-resident chain kernel (fp32 mode)
    int bf16_store = 1;    // FEAR_OPT_BF16_STORE: math 2 keeps the activations of the trunk's HBM-bound front in bf16 between kernels
    int chain32 = 2;       // FEAR_OPT_CHAIN32: 1 = the 32 x 32 trunk stage (four blocks) as one register-resident chain kernel (chain32_kernel; fp32 mode, throughput plan), 2 = in one launch with the stride-16 stage + neck (chain32_16_kernel) when FEAR_OPT_CHAIN is on as well
    int e1_pair = 1;       // FEAR_OPT_E1_PAIR: 1 = two consecutive 24-channel e1 blocks as one launch (e1pair_kernel; fp32 mode, throughput plan)
    int head_chain = 1;    // FEAR_OPT_HEAD_CHAIN: 1 = the whole BoxTower as one launch (headchain_kernel; fp32 mode, throughput plan)
    int small_pass = 96;   // passes of at most this many crops run the small-batch plan (FEAR_OPT_SMALL_PASS; 0: never);
                           // measured crossover with the throughput plan: ~110 crops (1.70 vs 1.93 ms at 96, 2.14 vs 2.01 at 128)
    int math = 0;          // 0: fp32 MFMA everywhere; 1: fp16-split operands on the matrix pipe in the fused 16x16 blocks
    bool fused_attr_set = false;
    int last_hip_error = 0;
    std::map<std::pair<int, int>, std::unique_ptr<Plan>> plans;
    float* workspace = nullptr;
    size_t workspace_floats = 0;
    std::vector<float*> weight_allocs;    // per-conv weights: live as long as the handle
    std::vector<float*> plan_allocs;      // weights packed for the fused kernels of the cached plans: freed with the plans
    bool building_plan = false;           // upload() books into plan_allocs while a plan is being built
    int head_stagger_us = 0;              // FEAR_OPT_HEAD_STAGGER: the second branch starts this many microseconds after the first
    int dual_head = 0;                    // FEAR_OPT_DUAL_HEAD: throughput plan runs the head's two branches on two streams (A/B option:
                                          // measured 104.3 k vs 104.8 k crops/s single-stream — the 16x16 kernels are ALU-bound, a second
                                          // resident workgroup per CU buys nothing: kbench 512 vs 2 x 256 crops, +2 %)
    int plan_crops = 0;                   // FEAR_OPT_PLAN_CROPS: crop count whose plan the introspection calls describe (0: max_batch)
    hipStream_t last_stream = nullptr;    // the caller stream of the previous call (compared only): a call on another stream first waits for `stream_switch`
    bool last_stream_valid = false;       // (the workspace and the branch stream are shared by all calls on a handle)
    hipEvent_t stream_switch = nullptr;
    std::vector<hipEvent_t> event_pool;   // recycled profiling events (creation is slow enough to perturb timing)
    hipStream_t branch_stream = nullptr;  // second stream for the bbox branch of the head at small batch sizes
    hipEvent_t branch_fork = nullptr, branch_join = nullptr;
    int split_streams = 0;                // FEAR_OPT_SPLIT_STREAMS: a throughput pass runs as two half-batches on two streams
    hipStream_t split_stream = nullptr;
    hipEvent_t split_fork = nullptr, split_join = nullptr;
};

namespace {

#define HIP_TRY(h, expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            (h)->last_hip_error = (int)_e;                 \
            return FEAR_ERR_HIP;                           \
        }                                                  \
    } while (0)

int upload(fear_handle* h, const std::vector<float>& host, float** dev) {
    *dev = nullptr;
    if (host.empty()) return FEAR_OK;
    float* p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(float)) != hipSuccess) return FEAR_ERR_ALLOC;
    (h->building_plan ? h->plan_allocs : h->weight_allocs).push_back(p);
    HIP_TRY(h, hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    *dev = p;
    return FEAR_OK;
}

int parse_model(fear_handle* h, const uint8_t* blob, size_t nbytes) {
    if (nbytes < sizeof(FearwHeader)) return FEAR_ERR_FORMAT;
    FearwHeader hd;
    memcpy(&hd, blob, sizeof(hd));
    if (memcmp(hd.magic, FEARW_MAGIC, 8) != 0 || hd.version != FEARW_VERSION || hd.payload_dtype > FEARW_PAYLOAD_F32)
        return FEAR_ERR_FORMAT;
    const size_t esz = hd.payload_dtype == FEARW_PAYLOAD_F32 ? 4 : 2;
    auto read_elems = [&](const uint8_t* src, size_t n, std::vector<float>& dst) {
        dst.resize(n);
        if (esz == 4) memcpy(dst.data(), src, n * 4);
        else
            for (size_t j = 0; j < n; ++j) { uint16_t v; memcpy(&v, src + 2 * j, 2); dst[j] = half_to_float(v); }
    };
    const size_t tables = sizeof(FearwHeader) + (size_t)hd.n_convs * sizeof(FearwConv) +
                          (size_t)hd.n_blocks * sizeof(FearwBlock);
    if (tables > nbytes || hd.payload_bytes > nbytes - tables) return FEAR_ERR_FORMAT;
    const uint8_t* payload = blob + tables;
    const FearwConv* ct = reinterpret_cast<const FearwConv*>(blob + sizeof(FearwHeader));
    const FearwBlock* bt = reinterpret_cast<const FearwBlock*>(blob + sizeof(FearwHeader) +
                                                               (size_t)hd.n_convs * sizeof(FearwConv));
    h->convs.resize(hd.n_convs);
    for (uint32_t i = 0; i < hd.n_convs; ++i) {
        FearwConv fc;
        memcpy(&fc, ct + i, sizeof(fc));
        Conv& c = h->convs[i];
        c.cout = fc.cout; c.cin_g = fc.cin_per_group; c.groups = fc.groups; c.k = fc.k;
        c.stride = fc.stride; c.pad = fc.pad; c.relu = fc.relu; c.has_bias = fc.has_bias;
        // bounded dimensions first, so that the products below cannot wrap and a malformed blob cannot ask for a huge allocation
        if (c.cout < 1 || c.cout > 8192 || c.cin_g < 1 || c.cin_g > 8192 || c.groups < 1 || c.groups > 8192 ||
            (c.k != 1 && c.k != 3 && c.k != 5))
            return FEAR_ERR_FORMAT;
        const size_t nw = (size_t)c.cout * c.cin_g * c.k * c.k;
        if (fc.w_off > hd.payload_bytes || nw * esz > hd.payload_bytes - fc.w_off) return FEAR_ERR_FORMAT;
        if (c.pad != c.k / 2 || (c.stride != 1 && c.stride != 2)) return FEAR_ERR_FORMAT;
        read_elems(payload + fc.w_off, nw, c.w);
        if (c.has_bias) {
            if (fc.b_off > hd.payload_bytes || (size_t)c.cout * esz > hd.payload_bytes - fc.b_off) return FEAR_ERR_FORMAT;
            read_elems(payload + fc.b_off, (size_t)c.cout, c.b);
        }
    }
    h->blocks.resize(hd.n_blocks);
    for (uint32_t i = 0; i < hd.n_blocks; ++i) {
        memcpy(&h->blocks[i], bt + i, sizeof(FearwBlock));
        const FearwBlock& b = h->blocks[i];
        for (int j = 0; j < 3; ++j)
            if (b.conv[j] < -1 || b.conv[j] >= (int)hd.n_convs) return FEAR_ERR_FORMAT;
        // mandatory conv slots per block kind: stem / neck conv[0]; IR depthwise + project (expand optional); SepConv dw + pw
        const bool ok = b.kind == FEARW_STEM || b.kind == FEARW_NECK ? b.conv[0] >= 0
                      : b.kind == FEARW_IR                           ? b.conv[1] >= 0 && b.conv[2] >= 0
                      : b.kind == FEARW_SEP                          ? b.conv[0] >= 0 && b.conv[1] >= 0
                                                                     : false;
        if (!ok) return FEAR_ERR_FORMAT;
        if (b.kind == FEARW_SEP) h->has_head = true;
        if (b.kind == FEARW_NECK) h->feat_channels = h->convs[b.conv[0]].cout;
    }
    if (h->blocks.empty() || h->blocks[0].kind != FEARW_STEM || h->feat_channels == 0) return FEAR_ERR_FORMAT;
    return FEAR_OK;
}

// Re-lay-out and upload weights: pointwise [N][K] as is; depthwise -> [k*k][C]; stem -> [27][16].
int pack_weights(fear_handle* h) {
    for (Conv& c : h->convs) {
        std::vector<float> packed;
        if (c.is_pw()) {
            if (c.cin_g % 4 != 0) return FEAR_ERR_FORMAT;
            packed = c.w;
        } else if (c.is_dw()) {
            if (c.cout % 4 != 0 || (c.k != 3 && c.k != 5)) return FEAR_ERR_FORMAT;
            const int kk = c.k * c.k;
            packed.resize((size_t)kk * c.cout);
            for (int ch = 0; ch < c.cout; ++ch)
                for (int t = 0; t < kk; ++t) packed[(size_t)t * c.cout + ch] = c.w[(size_t)ch * kk + t];
        } else {  // stem
            if (c.cout != 16 || c.cin_g != 3 || c.k != 3 || c.stride != 2 || !c.has_bias) return FEAR_ERR_FORMAT;
            packed.resize(27 * 16);
            for (int o = 0; o < 16; ++o)
                for (int t = 0; t < 27; ++t) packed[t * 16 + o] = c.w[o * 27 + t];
            std::vector<float> rows(16 * 32, 0.f);
            for (int o = 0; o < 16; ++o)
                for (int t = 0; t < 27; ++t) rows[o * 32 + t] = c.w[o * 27 + t];
            int st2 = upload(h, rows, &c.d_w2);
            if (st2 != FEAR_OK) return st2;
        }
        int st = upload(h, packed, &c.d_w);
        if (st != FEAR_OK) return st;
        st = upload(h, c.b, &c.d_b);
        if (st != FEAR_OK) return st;
    }
    return FEAR_OK;
}


// ---------------------------------------------------------------------------------------------
// Fused 16x16 block kernels (ir16_fused_kernel): one instantiation per (CIN, CEXP, COUT, KS, EXPAND).
struct Fused16 {
    int cin, cexp, cout, ks, expand;
    void (*kernel)(Ir2Args);
    int lds_bytes;
    void (*kernel_splitk)(Ir2Args);     // small-batch variant: several workgroups per crop, one chunk range each (or nullptr)
    int splitk_kc;                      // chunks per workgroup compiled into that variant (0: run-time, Ir2Args::kc_count)
    void (*kernel_tiny)(Ir2Args);       // 16-channel output slice of kTinyRows map rows per workgroup (sep16_tiny_kernel), a handful of crops
    int lds_tiny;
};
#define FUSED16(CIN, CEXP, COUT, KS, EXP) \
    {CIN, CEXP, COUT, KS, EXP, ir16v2_fused_kernel<CIN, CEXP, COUT, KS, (EXP) != 0>, Ir2Geom<CIN, CEXP, COUT, KS, (EXP) != 0>::LDS_BYTES, \
     ir16v2_fused_kernel<CIN, CEXP, COUT, KS, (EXP) != 0, true>, 0}
#define SEP16(CIN, COUT, KS, KC) \
    {CIN, CIN, COUT, KS, 0, sep16_kernel<CIN, COUT, KS>, Sep16Geom<CIN, COUT, KS>::LDS_BYTES, sep16_kernel<CIN, COUT, KS, false, false, KC>, KC}
#ifndef FEAR_TINY_ROWS
#define FEAR_TINY_ROWS 2
#endif
#ifndef FEAR_TINY_KSPLIT
#define FEAR_TINY_KSPLIT 4
#endif
// (batch-1 track call, ms, rows x wave groups: 8x1 0.375 | 4x1 0.357 | 4x2 0.352 | 2x2 0.344 | 2x4 0.342; sep16_kernel<CIN, 16, 3> 0.383)
constexpr int kTinyRows = FEAR_TINY_ROWS;       // map rows per workgroup of sep16_tiny_kernel (one wave each)
constexpr int kTinyKSplit = FEAR_TINY_KSPLIT;   // wave groups per workgroup, each over its share of the input chunks
#define SEP16T(CIN, KS, KC) \
    {CIN, CIN, 16, KS, 0, sep16_kernel<CIN, 16, KS>, Sep16Geom<CIN, 16, KS>::LDS_BYTES, sep16_kernel<CIN, 16, KS, false, false, KC>, KC, \
     sep16_tiny_kernel<CIN, KS, kTinyRows, kTinyKSplit>, Sep16TinyGeom<CIN, KS, kTinyRows, kTinyKSplit>::LDS_BYTES}
const Fused16 kFused16[] = {
    FUSED16(64, 192, 64, 5, 1),   FUSED16(64, 384, 64, 5, 1),  FUSED16(64, 384, 112, 5, 1),
    FUSED16(112, 672, 112, 5, 1), FUSED16(112, 336, 112, 5, 1),
    SEP16(256, 256, 3, 4), SEP16(320, 256, 3, 4),
    SEP16T(256, 3, 2),                 // bbox_pred / cls_pred: 4 / 1 output channels padded to one 16-channel tile
    SEP16T(320, 3, 2),                 // N-split slices of the 320 -> 256 SepConv (very small batches)
};

// Spatially tiled fused block kernels (ir_tile_fused_kernel) for the high-resolution trunk stages.
struct FusedTile {
    int cin, cexp, cout, ks, st, expand, hw;   // hw: input map size the tiling was chosen for
    int tw, th;
    void (*kernel)(IrT2Args);
    int lds_bytes;
    int nw;                                     // waves per workgroup
};
#define FTILE(CIN, CEXP, CEXPP, COUT, KS, ST, TW, TH, EXP, MINW, HW)                                             \
    {CIN, CEXP, COUT, KS, ST, EXP, HW, TW, TH, ir_tile_v2_kernel<CIN, CEXPP, COUT, KS, ST, TW, TH, (EXP) != 0, MINW>, \
     IrT2Geom<CIN, CEXPP, COUT, KS, ST, TW, TH, (EXP) != 0>::LDS_BYTES, 8}
const FusedTile kFusedTile[] = {
    FTILE(16, 16, 16, 16, 3, 1, 32, 16, 0, 4, 128),    // fbnet_c stage 1  (e1, 128x128)
    FTILE(16, 96, 96, 24, 3, 2, 16, 8, 1, 4, 128),     // stage 2          (e6 s2, 128 -> 64)
    FTILE(24, 24, 32, 24, 3, 1, 16, 16, 0, 4, 64),     // stages 4, 5      (e1, 64x64; 24 channels padded to 32)
    FTILE(24, 144, 144, 32, 5, 2, 16, 16, 1, 2, 64),   // stage 6          (e6 s2, 64 -> 32)
    FTILE(32, 96, 96, 32, 5, 1, 16, 16, 1, 2, 32),     // stage 7  (4 corner tiles: 18x18 clipped region)
    FTILE(32, 192, 192, 32, 5, 1, 16, 32, 1, 2, 32),   // stage 8  (two 16x32 tiles per map: 4 rows per wave share the depthwise reads)
    FTILE(32, 192, 192, 32, 3, 1, 32, 32, 1, 2, 32),   // stage 9  (the whole 32x32 map per workgroup: no halo to re-expand)
    FTILE(32, 192, 192, 64, 5, 2, 16, 16, 1, 2, 32),   // stage 10         (e6 s2, 32 -> 16: the whole 16x16 output map)
};
#define FTILEH(CIN, CEXP, CEXPP, COUT, KS, ST, TW, TH, EXP, NW, MINW, HW)                                             \
    {CIN, CEXP, COUT, KS, ST, EXP, HW, TW, TH,                                                                        \
     ir_tile_h_kernel<CIN, CEXPP, COUT, KS, ST, TW, TH, (EXP) != 0, NW, MINW>,                                         \
     IrTHGeom<CIN, CEXPP, COUT, KS, ST, TW, TH, (EXP) != 0, NW>::LDS_BYTES, NW}
// same blocks, fp16-split operands on the matrix pipe (FEAR_OPT_MATH = 1); tiles sized for the 144 B/pixel LDS tile
const FusedTile kFusedTileH[] = {
    FTILEH(16, 16, 32, 16, 3, 1, 32, 8, 0, 8, 2, 128),     // stage 1
    FTILEH(16, 96, 96, 24, 3, 2, 16, 4, 1, 4, 2, 128),     // stage 2
    FTILEH(24, 24, 32, 24, 3, 1, 16, 16, 0, 8, 2, 64),     // stages 4, 5
    FTILEH(24, 144, 160, 32, 5, 2, 16, 4, 1, 4, 2, 64),    // stage 6   (144 -> 160 padded channels)
    FTILEH(32, 96, 96, 32, 5, 1, 16, 16, 1, 8, 2, 32),     // stage 7
    FTILEH(32, 192, 192, 32, 5, 1, 16, 16, 1, 8, 2, 32),   // stage 8
    FTILEH(32, 192, 192, 32, 3, 1, 16, 16, 1, 8, 2, 32),   // stage 9
    FTILEH(32, 192, 192, 64, 5, 2, 16, 4, 1, 4, 2, 32),    // stage 10
};
static_assert(sizeof(kFusedTileH) == sizeof(kFusedTile), "the two tile tables must list the same blocks in the same order");
#define FTILEB(CIN, CEXP, CEXPP, COUT, KS, ST, TW, TH, EXP, NW, MINW, HW)                                             \
    {CIN, CEXP, COUT, KS, ST, EXP, HW, TW, TH,                                                                        \
     ir_tile_h_kernel<CIN, CEXPP, COUT, KS, ST, TW, TH, (EXP) != 0, NW, MINW, 2>,                                      \
     IrTHGeom<CIN, CEXPP, COUT, KS, ST, TW, TH, (EXP) != 0, NW>::LDS_BYTES, NW}
// same blocks, bf16 operands (FEAR_OPT_MATH = 2): one matrix-pipe MFMA per product instead of two
const FusedTile kFusedTileB[] = {
    FTILEB(16, 16, 32, 16, 3, 1, 32, 8, 0, 8, 2, 128),
    FTILEB(16, 96, 96, 24, 3, 2, 16, 4, 1, 4, 2, 128),
    FTILEB(24, 24, 32, 24, 3, 1, 16, 16, 0, 8, 2, 64),
    FTILEB(24, 144, 160, 32, 5, 2, 16, 4, 1, 4, 2, 64),
    FTILEB(32, 96, 96, 32, 5, 1, 16, 16, 1, 8, 2, 32),
    FTILEB(32, 192, 192, 32, 5, 1, 16, 16, 1, 8, 2, 32),
    FTILEB(32, 192, 192, 32, 3, 1, 16, 16, 1, 8, 2, 32),
    FTILEB(32, 192, 192, 64, 5, 2, 16, 4, 1, 4, 2, 32),
};
static_assert(sizeof(kFusedTileB) == sizeof(kFusedTile), "same blocks, same order");

// stem (3x3 s2, 3 -> 16) fused in front of the first e1 block: one entry, keyed by the block's shape and map size
const FusedTile kStemTile = {16, 16, 16, 3, 1, 0, 128, 32, 16,
                             ir_tile_v2_kernel<27, 16, 16, 3, 1, 32, 16, true, 4, true>,
                             IrT2Geom<27, 16, 16, 3, 1, 32, 16, true>::LDS_BYTES, 8};

// FEAR_OPT_MATH = 2 with FEAR_OPT_BF16_STORE: the HBM-bound front of the trunk keeps its activations in bf16 between kernels (stem
// output ... input of the 64 -> 32 block).  Storage variants of the tile kernels, keyed by (stem | index into the tile tables) and
// the IO bits (fear_kernels.h: IO_X_BF16 | IO_Y_BF16 | IO_R_BF16); same tiles and LDS footprints as the entries they stand in for.
struct TileBf16 { int stem, id, io; void (*kernel)(IrT2Args); };
const TileBf16 kTileBf16[] = {
    {1, -1, IO_Y_BF16, ir_tile_v2_kernel<27, 16, 16, 3, 1, 32, 16, true, 4, true, 0, IO_Y_BF16>},
    {0, 0, IO_X_BF16 | IO_Y_BF16 | IO_R_BF16, ir_tile_v2_kernel<16, 16, 16, 3, 1, 32, 16, false, 4, false, 0, IO_X_BF16 | IO_Y_BF16 | IO_R_BF16>},
    {0, 0, IO_X_BF16 | IO_Y_BF16, ir_tile_v2_kernel<16, 16, 16, 3, 1, 32, 16, false, 4, false, 0, IO_X_BF16 | IO_Y_BF16>},
    {0, 1, IO_X_BF16 | IO_Y_BF16, ir_tile_h_kernel<16, 96, 24, 3, 2, 16, 4, true, 4, 2, 2, IO_X_BF16 | IO_Y_BF16>},
    {0, 2, IO_X_BF16 | IO_Y_BF16 | IO_R_BF16, ir_tile_v2_kernel<24, 32, 24, 3, 1, 16, 16, false, 4, false, 0, IO_X_BF16 | IO_Y_BF16 | IO_R_BF16>},
    {0, 2, IO_X_BF16 | IO_Y_BF16, ir_tile_v2_kernel<24, 32, 24, 3, 1, 16, 16, false, 4, false, 0, IO_X_BF16 | IO_Y_BF16>},
    {0, 3, IO_X_BF16, ir_tile_h_kernel<24, 160, 32, 5, 2, 16, 4, true, 4, 2, 2, IO_X_BF16>},
};
const TileBf16* find_tile_bf16(int stem, int id, int io) {
    for (const TileBf16& t : kTileBf16)
        if (t.stem == stem && (stem || t.id == id) && t.io == io) return &t;
    return nullptr;
}

// the same blocks with smaller tiles = more workgroups per crop, for the small-batch plan (same order as kFusedTile)
const FusedTile kFusedTileSmall[] = {
    FTILE(16, 16, 16, 16, 3, 1, 32, 16, 0, 4, 128),
    FTILE(16, 96, 96, 24, 3, 2, 16, 8, 1, 4, 128),
    FTILE(24, 24, 32, 24, 3, 1, 16, 16, 0, 4, 64),
    FTILE(24, 144, 144, 32, 5, 2, 16, 8, 1, 4, 64),    // 8 tiles per crop instead of 4
    FTILE(32, 96, 96, 32, 5, 1, 16, 16, 1, 2, 32),
    FTILE(32, 192, 192, 32, 5, 1, 16, 16, 1, 2, 32),   // 4 instead of 2
    FTILE(32, 192, 192, 32, 3, 1, 16, 16, 1, 2, 32),   // 4 instead of 2
    FTILE(32, 192, 192, 64, 5, 2, 16, 8, 1, 2, 32),    // 2 instead of 1
};
static_assert(sizeof(kFusedTileSmall) == sizeof(kFusedTile), "same blocks, same order");
// a handful of crops (tiny plan): the smallest tiles the kernel supports, a tile's chunk loop is the latency that counts
const FusedTile kFusedTileTiny[] = {
    FTILE(16, 16, 16, 16, 3, 1, 32, 16, 0, 4, 128),
    FTILE(16, 96, 96, 24, 3, 2, 16, 8, 1, 4, 128),
    FTILE(24, 24, 32, 24, 3, 1, 16, 16, 0, 4, 64),
    FTILE(24, 144, 144, 32, 5, 2, 16, 8, 1, 4, 64),
    FTILE(32, 96, 96, 32, 5, 1, 16, 8, 1, 2, 32),      // 8 tiles per crop
    FTILE(32, 192, 192, 32, 5, 1, 16, 8, 1, 2, 32),    // 8
    FTILE(32, 192, 192, 32, 3, 1, 16, 8, 1, 2, 32),    // 8
    FTILE(32, 192, 192, 64, 5, 2, 16, 8, 1, 2, 32),
};
static_assert(sizeof(kFusedTileTiny) == sizeof(kFusedTile), "same blocks, same order");
// split-K variants of tiny-plan tiles (ir_tile_v2_kernel<..., KSPLIT>): the blocks whose chunk loop is the longest serial
// stretch of a one-crop pass get gridDim.y workgroups per tile + a splitk_reduce launch
struct TileKSplit {
    int id, k;                                  // index into kFusedTileTiny, chunks per workgroup
    void (*kernel)(IrT2Args);
    int lds_bytes;
};
#define TKSPLIT(ID, K, CIN, CEXP, COUT, KS, ST, TW, TH, MINW) \
    {ID, K, ir_tile_v2_kernel<CIN, CEXP, COUT, KS, ST, TW, TH, true, MINW, false, K>, IrT2Geom<CIN, CEXP, COUT, KS, ST, TW, TH, true>::LDS_BYTES}
const TileKSplit kTileKSplit[] = {
    TKSPLIT(3, 3, 24, 144, 32, 5, 2, 16, 8, 4),     // stage 6:  9 chunks -> 3 workgroups per tile
    TKSPLIT(4, 2, 32, 96, 32, 5, 1, 16, 8, 2),      // stage 7:  6 chunks -> 3
    TKSPLIT(5, 3, 32, 192, 32, 5, 1, 16, 8, 2),     // stage 8: 12 chunks -> 4
    TKSPLIT(6, 3, 32, 192, 32, 3, 1, 16, 8, 2),     // stage 9: 12 chunks -> 4
    TKSPLIT(7, 2, 32, 192, 64, 5, 2, 16, 8, 2),     // stage 10: 12 chunks -> 6 (two tiles per crop)
};
const TileKSplit* find_tile_ksplit(int id) {
    for (const TileKSplit& t : kTileKSplit)
        if (t.id == id) return &t;
    return nullptr;
}
// Throughput-plan blocks that run the phase-overlapped kernel (ir_tile_v4_kernel: same tile, packed weights, LDS bytes and
// arguments as their kFusedTile entry) — the tiles whose E tile leaves room for ONE workgroup per CU, where nobody else fills
// the LDS round trips of the depthwise; kernel = nullptr: the block stays on ir_tile_v2.  tools/kbench A/B, 256 crops
// (profiles/r03_tile_v4_kbench.txt): stage 6 +5.5 %; stages 7-10 and 2 lose 3-25 % (the overlap costs ~90 more VGPRs: stages
// 7 / 8 drop from 2-4 workgroups per CU to 1-2, stage 9's 8 rows per wave spill) and keep v2.
#define FTILE4(CIN, CEXP, COUT, KS, ST, TW, TH, MINW, HW)                                                        \
    {CIN, CEXP, COUT, KS, ST, 1, HW, TW, TH, ir_tile_v4_kernel<CIN, CEXP, COUT, KS, ST, TW, TH, MINW>,           \
     IrT4Geom<CIN, CEXP, COUT, KS, ST, TW, TH>::LDS_BYTES, 8}
const FusedTile kFusedTileV4[] = {
    {}, {}, {},
    FTILE4(24, 144, 32, 5, 2, 16, 16, 2, 64),          // stage 6
    {}, {}, {}, {},
};
static_assert(sizeof(kFusedTileV4) == sizeof(kFusedTile), "same blocks, same order");
const FusedTile& fp32_tile(int small_tiles, int id) {
    if (small_tiles == 3) return kFusedTileV4[id];
    return small_tiles == 2 ? kFusedTileTiny[id] : small_tiles ? kFusedTileSmall[id] : kFusedTile[id];
}

int find_fused_tile(int cin, int cexp, int cout, int ks, int st, int expand, int hw) {
    for (size_t i = 0; i < sizeof(kFusedTile) / sizeof(kFusedTile[0]); ++i) {
        const FusedTile& f = kFusedTile[i];
        if (f.cin == cin && f.cexp == cexp && f.cout == cout && f.ks == ks && f.st == st && f.expand == expand &&
            f.hw == hw)
            return (int)i;
    }
    return -1;
}

#define FUSED16H(CIN, CEXP, COUT, KS, EXP) \
    {CIN, CEXP, COUT, KS, EXP, ir16h_fused_kernel<CIN, CEXP, COUT, KS, (EXP) != 0>, IrHGeom<CIN, CEXP, COUT, KS, (EXP) != 0>::LDS_BYTES}
// same shapes as kFused16, fp16-split operands on the matrix pipe (FEAR_OPT_MATH = 1)
const Fused16 kFused16H[] = {
    FUSED16H(64, 192, 64, 5, 1),   FUSED16H(64, 384, 64, 5, 1),  FUSED16H(64, 384, 112, 5, 1),
    FUSED16H(112, 672, 112, 5, 1), FUSED16H(112, 336, 112, 5, 1),
    FUSED16H(256, 256, 256, 3, 0), FUSED16H(320, 320, 256, 3, 0),
    FUSED16H(256, 256, 16, 3, 0),
    FUSED16H(320, 320, 16, 3, 0),
};
static_assert(sizeof(kFused16H) == sizeof(kFused16), "the two tables must list the same shapes in the same order");
#define FUSED16B(CIN, CEXP, COUT, KS, EXP) \
    {CIN, CEXP, COUT, KS, EXP, ir16h_fused_kernel<CIN, CEXP, COUT, KS, (EXP) != 0, 2>, IrHGeom<CIN, CEXP, COUT, KS, (EXP) != 0>::LDS_BYTES}
// bf16 operands (FEAR_OPT_MATH = 2)
const Fused16 kFused16B[] = {
    FUSED16B(64, 192, 64, 5, 1),   FUSED16B(64, 384, 64, 5, 1),  FUSED16B(64, 384, 112, 5, 1),
    FUSED16B(112, 672, 112, 5, 1), FUSED16B(112, 336, 112, 5, 1),
    FUSED16B(256, 256, 256, 3, 0), FUSED16B(320, 320, 256, 3, 0),
    FUSED16B(256, 256, 16, 3, 0),
    FUSED16B(320, 320, 16, 3, 0),
};
static_assert(sizeof(kFused16B) == sizeof(kFused16), "same shapes, same order");

// last tower SepConv + the prediction SepConv in one kernel (fp32 mode)
auto* const kSep16PredKernel = sep16_kernel<256, 256, 3, false, true>;
constexpr int kSep16PredLds = Sep16Geom<256, 256, 3>::LDS_BYTES;


// encode SepConv + pixel-wise correlation in one kernel (fp32 mode)
constexpr int kCorrC = 256, kCorrTz = 64;
auto* const kSep16CorrKernel = sep16_kernel<kCorrC, kCorrC, 3, true>;
constexpr int kSep16CorrLds = Sep16Geom<kCorrC, kCorrC, 3, true>::LDS_BYTES;

int find_fused16(int cin, int cexp, int cout, int ks, int expand) {
    if (cout <= 4) cout = 16;     // prediction heads run on the one-tile instantiation
    for (size_t i = 0; i < sizeof(kFused16) / sizeof(kFused16[0]); ++i) {
        const Fused16& f = kFused16[i];
        if (f.cin == cin && f.cexp == cexp && f.cout == cout && f.ks == ks && f.expand == expand) return (int)i;
    }
    return -1;
}


// Pack the weights of one fused 16x16 block per 16-channel chunk in the order ir16v2_fused_kernel stages them:
//   [A-part: KG fragments x 256 | be[16]] [BC-part: NTP fragments x 256 | Wd[k*k][16] | bd[16]]
// (fragment lane l holds W[row0 + (l&15)][col0 + 4*(l>>4) + 0..3]).
std::vector<float> pack_fused16_host(const fear_handle* h, int ce, int cd, int cp) {
    const Conv& d = h->convs[cd];
    const Conv& p = h->convs[cp];
    const Conv* e = ce >= 0 ? &h->convs[ce] : nullptr;
    const int cexp = d.cout, cout = p.cout, kk = d.k * d.k;
    const int cin = e ? e->cin_g * e->k * e->k : cexp;   // 27 for the stem conv used as the 'expand' of a fused stem tile
    const int kg_n = e ? (cin + 15) / 16 : 0, ntp = (cout + 15) / 16;
    const int cexpp = (cexp + 15) / 16 * 16;       // channels / rows beyond the real extent are packed as zeros
    std::vector<float> buf;
    for (int c0 = 0; c0 < cexpp; c0 += 16) {
        if (e) {
            for (int kg = 0; kg < kg_n; ++kg) {
                // a last k-group of 8 channels is packed two per lane group (k = 2*(l>>4) + i, i < 2): the tile kernel then
                // spends two MFMA steps on it instead of four (IrT2Geom::KHALF)
                const bool khalf = cin % 16 == 8 && kg == kg_n - 1;
                // the stem's im2col K = 27: MFMA step j = kg * 4 + i takes k = 4 * j + (l >> 4), so that the padding (k = 27..31) is
                // one lane group of step 6 and the whole of step 7, which the stem tile never issues (7 MFMAs per m-tile, not 8)
                const bool stemk = e->k > 1 && cin == 27;
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const int n = c0 + (l & 15);
                        const int k = stemk ? 4 * (kg * 4 + i) + (l >> 4) :
                                      khalf ? (i < 2 ? kg * 16 + (l >> 4) * 2 + i : cin) : kg * 16 + (l >> 4) * 4 + i;
                        buf.push_back(n < cexp && k < cin ? e->w[(size_t)n * cin + k] : 0.f);
                    }
            }
            for (int ch = 0; ch < 16; ++ch) buf.push_back(e->has_bias && c0 + ch < cexp ? e->b[c0 + ch] : 0.f);
        }
        for (int nt = 0; nt < ntp; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) {
                    const int n = nt * 16 + (l & 15), k = c0 + (l >> 4) * 4 + i;
                    buf.push_back(n < cout && k < cexp ? p.w[(size_t)n * cexp + k] : 0.f);
                }
        for (int t = 0; t < kk; ++t)
            for (int ch = 0; ch < 16; ++ch) buf.push_back(c0 + ch < cexp ? d.w[(size_t)(c0 + ch) * kk + t] : 0.f);
        for (int ch = 0; ch < 16; ++ch) buf.push_back(d.has_bias && c0 + ch < cexp ? d.b[c0 + ch] : 0.f);
    }
    return buf;
}

int pack_fused16(fear_handle* h, int ce, int cd, int cp, float** out) { return upload(h, pack_fused16_host(h, ce, cd, cp), out); }


// N-split packing of a SepConv (Ir2Args::nsplit_wstride): one weight set per 16-channel output slice, each in the layout
// sep16_kernel<CIN, 16, KS> stages — per 16-channel input chunk [1 projection fragment | Wd[k*k][16] | bd[16]].
int pack_sep16_nsplit(fear_handle* h, int cd, int cp, float** out) {
    const Conv& d = h->convs[cd];
    const Conv& p = h->convs[cp];
    const int cin = d.cout, cout = p.cout, kk = d.k * d.k;
    std::vector<float> buf;
    for (int n0 = 0; n0 < cout; n0 += 16)
        for (int c0 = 0; c0 < cin; c0 += 16) {
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) buf.push_back(p.w[(size_t)(n0 + (l & 15)) * cin + c0 + (l >> 4) * 4 + i]);
            for (int t = 0; t < kk; ++t)
                for (int ch = 0; ch < 16; ++ch) buf.push_back(d.w[(size_t)(c0 + ch) * kk + t]);
            for (int ch = 0; ch < 16; ++ch) buf.push_back(d.has_bias ? d.b[c0 + ch] : 0.f);
        }
    return upload(h, buf, out);
}

// Packed weights for the matrix-pipe (fp16-split) fused kernels, per 32-channel chunk (IrHGeom layout):
//   [A-part: 2 n-tiles x KG32 fragments of 64 lanes x 8 halfs | be[32] fp32]
//   [BC-part: NTP fragments | Wd[k*k][32] fp32 | bd[32] fp32]
// The model's weights are fp16 values, so the fp32 -> fp16 conversion here is exact.  bf16 = true (FEAR_OPT_MATH = 2)
// rounds them to bf16 instead.
int pack_fused_h(fear_handle* h, int ce, int cd, int cp, float** out, bool bf16 = false) {
    const Conv& d = h->convs[cd];
    const Conv& p = h->convs[cp];
    const Conv* e = ce >= 0 ? &h->convs[ce] : nullptr;
    const int cexp = d.cout, cout = p.cout, kk = d.k * d.k;
    const int cin = e ? e->cin_g : cexp;
    const int kg_n = e ? (cin + 31) / 32 : 0, ntp = (cout + 15) / 16;
    const int cexpp = (cexp + 31) / 32 * 32;
    std::vector<float> buf;
    auto push_frag = [&](const std::vector<float>& w, int rows, int cols, int row0, int col0) {
        for (int l = 0; l < 64; ++l) {
            uint16_t hv[8];
            for (int j = 0; j < 8; ++j) {
                const int r = row0 + (l & 15), c = col0 + (l >> 4) * 8 + j;
                const float wv = r < rows && c < cols ? w[(size_t)r * cols + c] : 0.f;
                hv[j] = bf16 ? float_to_bf16(wv) : float_to_half(wv);   // fp16 -> fp16 is exact; -> bf16 rounds to 8 bits
            }
            float f4[4];
            memcpy(f4, hv, 16);
            buf.insert(buf.end(), f4, f4 + 4);
        }
    };
    for (int c0 = 0; c0 < cexpp; c0 += 32) {
        if (e) {
            for (int nt = 0; nt < 2; ++nt)
                for (int kg = 0; kg < kg_n; ++kg) push_frag(e->w, cexp, cin, c0 + nt * 16, kg * 32);
            for (int ch = 0; ch < 32; ++ch) buf.push_back(e->has_bias && c0 + ch < cexp ? e->b[c0 + ch] : 0.f);
        }
        for (int nt = 0; nt < ntp; ++nt) push_frag(p.w, cout, cexp, nt * 16, c0);
        for (int t = 0; t < kk; ++t)
            for (int ch = 0; ch < 32; ++ch) buf.push_back(c0 + ch < cexp ? d.w[(size_t)(c0 + ch) * kk + t] : 0.f);
        for (int ch = 0; ch < 32; ++ch) buf.push_back(d.has_bias && c0 + ch < cexp ? d.b[c0 + ch] : 0.f);
    }
    return upload(h, buf, out);
}


// ---------------------------------------------------------------------------------------------
// The stride-16 trunk stage of FEAR-XS (fbnet_c stages 11-17 + AdjustLayer) as one register-resident chain kernel.
struct ChainShape { int cin, cexp, cout, ks, res; };
const ChainShape kChainXS[7] = {{64, 192, 64, 5, 1},   {64, 384, 64, 5, 1},   {64, 384, 64, 5, 1}, {64, 384, 112, 5, 0},
                                {112, 672, 112, 5, 1}, {112, 672, 112, 5, 1}, {112, 336, 112, 5, 1}};
constexpr int kChainNeckOut = 256;
auto* const kChainXSKernel =
    chain16_kernel<ChainBlk<64, 192, 64, 5, true>, ChainBlk<64, 384, 64, 5, true>, ChainBlk<64, 384, 64, 5, true>,
                   ChainBlk<64, 384, 112, 5, false>, ChainBlk<112, 672, 112, 5, true>, ChainBlk<112, 672, 112, 5, true>,
                   ChainBlk<112, 336, 112, 5, true>, kChainNeckOut>;
constexpr int kChainXSLds =
    Chain16Lds<5, Ir2Geom<112, 672, 112, 5, true>::AP, Ir2Geom<112, 672, 112, 5, true>::BP>::FLOATS * 4;

// The 32 x 32 trunk stage of FEAR-XS (fbnet_c stage 3 + the stride-2 block that opens stage 4) as one register-resident chain
// kernel (fear_chain32.h).
struct Chain32Shape { int cin, cexp, cout, ks, stride, res; };
const Chain32Shape kChain32XS[4] = {{32, 96, 32, 5, 1, 1}, {32, 192, 32, 5, 1, 1}, {32, 192, 32, 3, 1, 1}, {32, 192, 64, 5, 2, 0}};
auto* const kChain32XSKernel = chain32_kernel<C32Blk<32, 96, 32, 5, 1, true>, C32Blk<32, 192, 32, 5, 1, true>,
                                              C32Blk<32, 192, 32, 3, 1, true>, C32Blk<32, 192, 64, 5, 2, false>>;
auto* const kChain32_16XSKernel =
    chain32_16_kernel<C32Blk<32, 96, 32, 5, 1, true>, C32Blk<32, 192, 32, 5, 1, true>, C32Blk<32, 192, 32, 3, 1, true>, C32Blk<32, 192, 64, 5, 2, false>,
                      ChainBlk<64, 192, 64, 5, true>, ChainBlk<64, 384, 64, 5, true>, ChainBlk<64, 384, 64, 5, true>,
                      ChainBlk<64, 384, 112, 5, false>, ChainBlk<112, 672, 112, 5, true>, ChainBlk<112, 672, 112, 5, true>,
                      ChainBlk<112, 336, 112, 5, true>, kChainNeckOut>;
constexpr int kChain32_16Lds = kChainXSLds > C32Geom::LDS_BYTES ? kChainXSLds : C32Geom::LDS_BYTES;

// the whole BoxTower as one launch (fear_headchain.h): 3x3 SepConvs, 256 channels, 64 template positions
auto* const kHeadChainKernel = headchain_kernel<3>;
using HeadChainG = HeadChainGeom<3>;
auto* const kHeadChainBKernel = headchain_b_kernel<3>;      // FEAR_OPT_MATH = 2
using HeadChainBG = HeadChainBGeom<3>;

// neck weights as MFMA fragments [nt][kg][lane][4] (lane l: W[nt*16 + (l&15)][kg*16 + 4*(l>>4) + 0..3])
int pack_neck_frags(fear_handle* h, int conv, float** out) {
    const Conv& c = h->convs[conv];
    const int n = c.cout, k = c.cin_g;
    std::vector<float> buf;
    for (int nt = 0; nt < n / 16; ++nt)
        for (int kg = 0; kg < k / 16; ++kg)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) buf.push_back(c.w[(size_t)(nt * 16 + (l & 15)) * k + kg * 16 + (l >> 4) * 4 + i]);
    return upload(h, buf, out);
}

// ---------------------------------------------------------------------------------------------
// Plan construction.  Activation buffers come from a small pool of equally sized slabs
// (per-crop size = the largest intermediate tensor) handed out by liveness, so consecutive layers
// keep re-using the same few address ranges (friendly to the 256 MiB Infinity Cache).
struct Pool {
    std::vector<int> free_ids, held;
    int count = 0;
    bool hold = false;       // while set, released buffers are not handed out again (ops that may run concurrently)
    int acquire() {
        if (!free_ids.empty()) { int id = free_ids.back(); free_ids.pop_back(); return id; }
        return count++;
    }
    void release(int id) {
        if (id < 0) return;
        (hold ? held : free_ids).push_back(id);
    }
    void flush() {
        hold = false;
        free_ids.insert(free_ids.end(), held.begin(), held.end());
        held.clear();
    }
};

struct T {  // tensor view inside the plan
    int buf = -1, ld = 0, off = 0, C = 0, H = 0, W = 0;
};

void set_name(Op& op, const char* fmt, int a, int b, int c) { snprintf(op.name, sizeof(op.name), fmt, a, b, c); }

// small = the small-batch plan (few crops per pass): split-K 16x16 kernels, the head's branches on two streams
int build_plan_uncached(fear_handle* h, int hw, bool with_head, int mode, Plan** out);

// mode: 0 = throughput plan, 1 = small-batch plan, 2 = the small-batch plan with the deepest split (a handful of crops)
int build_plan(fear_handle* h, int hw, bool with_head, int mode, Plan** out) {
    auto it = h->plans.find(std::make_pair(hw, (with_head ? 1 : 0) + 2 * mode));
    if (it != h->plans.end()) { *out = it->second.get(); return FEAR_OK; }
    h->building_plan = true;          // packed weights uploaded from here on belong to the plan cache
    const int st = build_plan_uncached(h, hw, with_head, mode, out);
    h->building_plan = false;
    return st;
}

int build_plan_uncached(fear_handle* h, int hw, bool with_head, int mode, Plan** out) {
    const bool small = mode == 1 || mode == 2;      // (mode 3: the throughput plan of a mid-size pass — no 32 x 32 chain, see plan_mode)
    // chunks of a 16x16 block are dealt to at most this many workgroups per crop: every workgroup writes a full partial
    // output map, so a deep split only pays while the crops are few (batch 1: 0.545 -> 0.512 ms per track call)
    const int splitk_max = mode == 2 ? 24 : 8;
    auto key = std::make_pair(hw, (with_head ? 1 : 0) + 2 * mode);
    if (hw < 32 || hw % 32 != 0 || hw > 1024) return FEAR_ERR_SHAPE;
    if (with_head && !h->has_head) return FEAR_ERR_NOHEAD;
    std::unique_ptr<Plan> plan(new Plan);
    plan->hw = hw;
    plan->with_head = with_head;
    plan->small = small;
    Pool pool;
    size_t max_elems = 0;
    auto track = [&](const T& t) {
        size_t e = (size_t)t.H * t.W * t.ld;
        if (e > max_elems) max_elems = e;
    };
    std::vector<Op>& ops = plan->ops;

    auto add_pw = [&](int conv, const T& in, T& outT, const T* res, int relu, int out_ld, int out_off, int out_buf) {
        const Conv& c = h->convs[conv];
        Op op{};
        op.type = OP_PW; op.conv = conv;
        op.in_buf = in.buf; op.in_ld = in.ld; op.in_off = in.off;
        op.H = in.H; op.W = in.W; op.Ho = in.H; op.Wo = in.W; op.C = c.cin_g; op.N = c.cout;
        op.relu = relu;
        outT.buf = out_buf >= 0 ? out_buf : pool.acquire();
        outT.ld = out_ld > 0 ? out_ld : c.cout; outT.off = out_off; outT.C = c.cout; outT.H = in.H; outT.W = in.W;
        op.out_buf = outT.buf; op.out_ld = outT.ld; op.out_off = outT.off;
        if (res) { op.res_buf = res->buf; op.res_ld = res->ld; }
        set_name(op, "pw_%dx%d_hw%d", c.cin_g, c.cout, in.H);
        const double px = (double)in.H * in.W;
        op.flops = 2.0 * px * c.cin_g * c.cout;
        op.bytes = 4.0 * px * (c.cin_g + c.cout + (res ? c.cout : 0));
        ops.push_back(op);
        track(outT);
    };
    auto add_dw = [&](int conv, const T& in, T& outT, int relu) {
        const Conv& c = h->convs[conv];
        Op op{};
        op.type = OP_DW; op.conv = conv;
        op.in_buf = in.buf; op.in_ld = in.ld; op.in_off = in.off;
        op.H = in.H; op.W = in.W; op.Ho = in.H / c.stride; op.Wo = in.W / c.stride; op.C = c.cout; op.N = c.cout;
        op.relu = relu;
        outT.buf = pool.acquire(); outT.ld = c.cout; outT.off = 0; outT.C = c.cout; outT.H = op.Ho; outT.W = op.Wo;
        op.out_buf = outT.buf; op.out_ld = outT.ld; op.out_off = 0;
        set_name(op, "dw%d_c%d_hw%d", c.k * 10 + c.stride, c.cout, in.H);
        op.flops = 2.0 * op.Ho * op.Wo * c.cout * c.k * c.k;
        op.bytes = 4.0 * c.cout * ((double)in.H * in.W + (double)op.Ho * op.Wo);
        ops.push_back(op);
        track(outT);
    };

    // fused 16x16 block (returns false when no instantiation matches -> caller emits the unfused ops)
    auto add_fused16 = [&](int ce, int cd, int cp, const T& in, T& outT, const T* res, int relu_dw, int relu_out,
                           int out_ld, const char* tag) -> bool {
        if (!h->fuse || in.H != 16 || in.W != 16) return false;
        const Conv& d = h->convs[cd];
        const Conv& p = h->convs[cp];
        if (d.stride != 1) return false;
        const int cin = ce >= 0 ? h->convs[ce].cin_g : d.cout;
        const int id = find_fused16(cin, d.cout, p.cout, d.k, ce >= 0 ? 1 : 0);
        if (id < 0) return false;
        if (!p.has_bias) return false;
        Op op{};
        op.type = OP_IR16; op.fused_id = id; op.conv_e = ce; op.conv_d = cd; op.conv_p = cp;
        op.math = h->math;
        // a handful of crops: a SepConv runs as COUT/16 independent 16-channel output slices, one workgroup each, every one
        // with the whole (cheap) depthwise — finished outputs from one launch instead of split-K partials + a reduce launch
        const int id16 = (mode == 2 && !h->math && ce < 0 && p.cout > 16 && p.cout % 16 == 0 && d.cout % 16 == 0)
                             ? find_fused16(cin, d.cout, 16, d.k, 0) : -1;
        if (id16 >= 0) {
            op.nsplit = p.cout / 16;
            op.fused_id = id16;
            op.tiny = h->tiny_sep && kFused16[id16].kernel_tiny ? 1 : 0;
            if (pack_sep16_nsplit(h, cd, cp, &op.d_packed) != FEAR_OK) return false;
        } else if ((h->math ? pack_fused_h(h, ce, cd, cp, &op.d_packed, h->math == 2) : pack_fused16(h, ce, cd, cp, &op.d_packed)) != FEAR_OK)
            return false;
        op.in_buf = in.buf; op.in_ld = in.ld; op.in_off = in.off;
        op.H = 16; op.W = 16; op.Ho = 16; op.Wo = 16; op.C = cin; op.N = p.cout;
        op.relu_dw = relu_dw; op.relu = relu_out;
        outT.buf = pool.acquire(); outT.ld = out_ld > 0 ? out_ld : p.cout; outT.off = 0; outT.C = p.cout; outT.H = 16; outT.W = 16;
        op.out_buf = outT.buf; op.out_ld = outT.ld;
        if (res) { op.res_buf = res->buf; op.res_ld = res->ld; }
        // small passes: one workgroup per crop leaves the GPU idle — split the expansion chunks of a crop over several
        // workgroups (each projects its own chunks), then add the partial projections up
        if (!h->math && small && !op.nsplit && kFused16[id].kernel_splitk) {
            const int nchunk = d.cout / 16;
            int w = 0;
            if (kFused16[id].splitk_kc > 0) {
                w = nchunk / kFused16[id].splitk_kc;              // the variant's chunk count per workgroup is compiled in
            } else {
                for (int cand = splitk_max; cand >= 2 && !w; --cand)
                    if (nchunk % cand == 0 && nchunk / cand >= (mode == 2 ? 1 : 2)) w = cand;
            }
            if (w) {
                op.splitk = w;
                if ((size_t)w * 256 * p.cout > max_elems) max_elems = (size_t)w * 256 * p.cout;
                op.part_buf = pool.acquire();
                pool.release(op.part_buf);          // only alive inside this op (the next acquire may reuse it)
            }
        }
        snprintf(op.name, sizeof(op.name), op.splitk ? "%s_splitk_%dx%dx%d_k%d" : op.nsplit ? "%s_nsplit_%dx%dx%d_k%d" : "%s_%dx%dx%d_k%d", tag, cin,
                 d.cout, p.cout, d.k);
        op.flops = 2.0 * 256 * ((ce >= 0 ? (double)cin * d.cout : 0.0) + (double)d.cout * d.k * d.k + (double)d.cout * p.cout);
        op.bytes = 4.0 * 256 * (cin + p.cout + (res ? p.cout : 0));
        ops.push_back(op);
        track(outT);
        return true;
    };

    // prediction head (dw3x3 + 1x1 to 4 / 1 channels [+ exp]) as a fused 16x16 block writing the caller's NCHW map
    auto add_fused_pred = [&](int cd, int cp, const T& in, int act, int ext) -> bool {
        if (!h->fuse || in.H != 16 || in.W != 16) return false;
        const Conv& d = h->convs[cd];
        const Conv& p = h->convs[cp];
        if (d.stride != 1 || p.cout > 4 || !p.has_bias) return false;
        const int id = find_fused16(d.cout, d.cout, p.cout, d.k, 0);
        if (id < 0) return false;
        Op op{};
        op.type = OP_IR16; op.fused_id = id; op.conv_e = -1; op.conv_d = cd; op.conv_p = cp;
        op.math = h->math;
        if ((h->math ? pack_fused_h(h, -1, cd, cp, &op.d_packed, h->math == 2) : pack_fused16(h, -1, cd, cp, &op.d_packed)) != FEAR_OK)
            return false;
        op.in_buf = in.buf; op.in_ld = in.ld; op.in_off = in.off;
        op.H = 16; op.W = 16; op.Ho = 16; op.Wo = 16; op.C = d.cout; op.N = p.cout;
        op.pred_cout = p.cout; op.act = act; op.out_external = ext;
        op.tiny = mode == 2 && !h->math && h->tiny_sep && kFused16[id].kernel_tiny ? 1 : 0;
        snprintf(op.name, sizeof(op.name), "%s_%dx%d_k%d", ext == 3 ? "cls_pred16" : "bbox_pred16", d.cout, p.cout, d.k);
        op.flops = 2.0 * 256 * ((double)d.cout * d.k * d.k + (double)d.cout * p.cout);
        op.bytes = 4.0 * 256 * (d.cout + p.cout);
        ops.push_back(op);
        return true;
    };

    auto add_fused_tile = [&](int ce, int cd, int cp, const T& in, T& outT, const T* res) -> bool {
        if (!h->fuse || in.H != in.W) return false;
        const Conv& d = h->convs[cd];
        const Conv& p = h->convs[cp];
        const int cin = ce >= 0 ? h->convs[ce].cin_g : d.cout;
        const int id = find_fused_tile(cin, d.cout, p.cout, d.k, d.stride, ce >= 0 ? 1 : 0, in.H);
        if (id < 0) return false;
        const int use_h = ce >= 0 ? h->math : 0;   // e1 blocks (no expand GEMM) stay on the fp32 kernel
        // 1 / 2: the small-batch / tiny plans' tiles; 3: the throughput plan's phase-overlapped kernel where the table has one
        const int small_tiles = small && !use_h ? (mode == 2 ? 2 : 1) : (!small && !use_h && h->tile_v4 && kFusedTileV4[id].kernel ? 3 : 0);
        const FusedTile& f = use_h == 2 ? kFusedTileB[id] : use_h ? kFusedTileH[id] : fp32_tile(small_tiles, id);
        const int ho = in.H / d.stride;
        if (ho % f.th != 0 || ho % f.tw != 0) return false;
        if (!p.has_bias) return false;
        Op op{};
        op.type = OP_IRTILE; op.fused_id = id; op.conv_e = ce; op.conv_d = cd; op.conv_p = cp;
        op.math = use_h;
        op.small_tiles = small_tiles;
        if ((use_h ? pack_fused_h(h, ce, cd, cp, &op.d_packed, use_h == 2) : pack_fused16(h, ce, cd, cp, &op.d_packed)) != FEAR_OK)
            return false;
        op.in_buf = in.buf; op.in_ld = in.ld; op.in_off = in.off;
        op.H = in.H; op.W = in.W; op.Ho = ho; op.Wo = ho; op.C = cin; op.N = p.cout;
        op.relu_dw = 1; op.relu = 0;
        outT.buf = pool.acquire(); outT.ld = p.cout; outT.off = 0; outT.C = p.cout; outT.H = ho; outT.W = ho;
        op.out_buf = outT.buf; op.out_ld = outT.ld;
        if (res) { op.res_buf = res->buf; op.res_ld = res->ld; }
        if (small_tiles == 2 && ce >= 0) {
            if (const TileKSplit* ks = find_tile_ksplit(id)) {
                op.splitk = d.cout / 16 / ks->k;
                const size_t pe = (size_t)op.splitk * ho * ho * p.cout;
                if (pe > max_elems) max_elems = pe;
                op.part_buf = pool.acquire();
                pool.release(op.part_buf);          // only alive inside this op
            }
        }
        snprintf(op.name, sizeof(op.name), op.splitk ? "irt_splitk_%dx%dx%d_k%ds%d_hw%d" : "irt_%dx%dx%d_k%ds%d_hw%d", cin, d.cout, p.cout, d.k,
                 d.stride, in.H);
        op.flops = 2.0 * ((ce >= 0 ? (double)in.H * in.W * cin * d.cout : 0.0) +
                          (double)ho * ho * d.cout * d.k * d.k + (double)ho * ho * d.cout * p.cout);
        op.bytes = 4.0 * ((double)in.H * in.W * cin + (double)ho * ho * p.cout * (res ? 2 : 1));
        ops.push_back(op);
        track(outT);
        return true;
    };

    T cur;
    size_t bi = 0;
    // ---- stem fused with the first inverted-residual block (e1, residual) when the shapes match the instantiation
    bool stem_fused = false;
    if (h->fuse && h->blocks.size() > 1 && h->blocks[1].kind == FEARW_IR && h->blocks[1].conv[0] < 0 && h->blocks[1].residual) {
        const Conv& cs = h->convs[h->blocks[0].conv[0]];
        const Conv& d = h->convs[h->blocks[1].conv[1]];
        const Conv& p = h->convs[h->blocks[1].conv[2]];
        const FusedTile& f = kStemTile;
        const int ho = hw / 2;
        if (cs.cout == 16 && d.cout == f.cexp && p.cout == f.cout && d.k == f.ks && d.stride == 1 && ho == f.hw &&
            p.has_bias && d.relu && !p.relu) {
            Op op{};
            op.type = OP_IRTILE; op.fused_id = -1; op.stem = 1;
            op.conv_e = h->blocks[0].conv[0]; op.conv_d = h->blocks[1].conv[1]; op.conv_p = h->blocks[1].conv[2];
            if (pack_fused16(h, op.conv_e, op.conv_d, op.conv_p, &op.d_packed) == FEAR_OK) {
                op.H = ho; op.W = ho; op.Ho = ho; op.Wo = ho; op.C = 3; op.N = p.cout;
                op.relu_dw = 1; op.relu = 0;
                cur.buf = pool.acquire(); cur.ld = p.cout; cur.off = 0; cur.C = p.cout; cur.H = ho; cur.W = ho;
                op.out_buf = cur.buf; op.out_ld = cur.ld;
                snprintf(op.name, sizeof(op.name), "stem_irt_3x16x16_k3_hw%d", hw);
                op.flops = 2.0 * ho * ho * (27.0 * 16 + 16.0 * 9 + 16.0 * 16);
                op.bytes = 4.0 * (3.0 * hw * hw + (double)ho * ho * 16);
                ops.push_back(op);
                track(cur);
                stem_fused = true;
                bi = 2;
            }
        }
    }
    // ---- stem
    if (!stem_fused) {
        const FearwBlock& b = h->blocks[0];
        const Conv& c = h->convs[b.conv[0]];
        Op op{};
        op.type = OP_STEM; op.conv = b.conv[0];
        op.H = hw; op.W = hw; op.Ho = hw / 2; op.Wo = hw / 2; op.C = 3; op.N = c.cout; op.relu = 1;
        cur.buf = pool.acquire(); cur.ld = c.cout; cur.off = 0; cur.C = c.cout; cur.H = hw / 2; cur.W = hw / 2;
        op.out_buf = cur.buf; op.out_ld = cur.ld;
        set_name(op, "stem_3x%d_hw%d%.0d", c.cout, hw, 0);
        op.flops = 2.0 * op.Ho * op.Wo * 27 * c.cout;
        op.bytes = 4.0 * (3.0 * hw * hw + (double)op.Ho * op.Wo * c.cout);
        ops.push_back(op);
        track(cur);
        bi = 1;
    }
    // ---- trunk + neck
    for (; bi < h->blocks.size(); ++bi) {
        const FearwBlock& b = h->blocks[bi];
        // ---- the whole stride-16 stage + neck as one chain kernel (fp32 arithmetic, search branch)
        if (h->fuse && h->chain && !h->math && !small && with_head && b.kind == FEARW_IR && cur.H == 16 && cur.W == 16 &&
            bi + 7 < h->blocks.size() && h->blocks[bi + 7].kind == FEARW_NECK) {
            bool match = true;
            for (int j = 0; j < 7 && match; ++j) {
                const FearwBlock& bj = h->blocks[bi + j];
                if (bj.kind != FEARW_IR || bj.conv[0] < 0) { match = false; break; }
                const Conv& e = h->convs[bj.conv[0]];
                const Conv& d = h->convs[bj.conv[1]];
                const Conv& p = h->convs[bj.conv[2]];
                const ChainShape& cs = kChainXS[j];
                match = e.cin_g == cs.cin && d.cout == cs.cexp && p.cout == cs.cout && d.k == cs.ks && d.stride == 1 &&
                        (int)bj.residual == cs.res && e.has_bias && p.has_bias && e.relu && d.relu && !p.relu;
            }
            const Conv& nk = h->convs[h->blocks[bi + 7].conv[0]];
            match = match && nk.cin_g == kChainXS[6].cout && nk.cout == kChainNeckOut && nk.has_bias && !nk.relu;
            if (match) {
                Op op{};
                op.type = OP_CHAIN16;
                bool ok = true;
                double fl = 0;
                for (int j = 0; j < 7 && ok; ++j) {
                    const FearwBlock& bj = h->blocks[bi + j];
                    ok = pack_fused16(h, bj.conv[0], bj.conv[1], bj.conv[2], &op.chain_pk[j]) == FEAR_OK;
                    op.chain_cp[j] = bj.conv[2];
                    const ChainShape& cs = kChainXS[j];
                    fl += 2.0 * 256 * ((double)cs.cin * cs.cexp + 25.0 * cs.cexp + (double)cs.cexp * cs.cout);
                }
                op.neck_conv = h->blocks[bi + 7].conv[0];
                ok = ok && pack_neck_frags(h, op.neck_conv, &op.neck_pk) == FEAR_OK;
                if (ok) {
                    op.in_buf = cur.buf; op.in_ld = cur.ld; op.in_off = cur.off;
                    op.H = 16; op.W = 16; op.Ho = 16; op.Wo = 16; op.C = kChainXS[0].cin; op.N = kChainNeckOut;
                    T o;
                    o.buf = pool.acquire(); o.ld = kChainNeckOut; o.off = 0; o.C = kChainNeckOut; o.H = 16; o.W = 16;
                    op.out_buf = o.buf; op.out_ld = o.ld;
                    snprintf(op.name, sizeof(op.name), "chain16_7blocks_neck_%dx%d", op.C, op.N);
                    op.flops = fl + 2.0 * 256 * kChainXS[6].cout * kChainNeckOut;
                    op.bytes = 4.0 * 256 * (op.C + op.N);
                    if (h->chain32 == 2 && !ops.empty() && ops.back().type == OP_CHAIN32 && ops.back().out_buf == op.in_buf && op.in_off == 0) {
                        // the 32 x 32 stage in front of it is a chain kernel too: one launch for both (chain32_16_kernel), the
                        // 16 x 16 x 64 map between them stays in registers
                        Op& f = ops.back();
                        f.type = OP_CHAIN32_16;
                        for (int j = 0; j < 7; ++j) { f.chain16_pk[j] = op.chain_pk[j]; f.chain16_cp[j] = op.chain_cp[j]; }
                        f.neck_conv = op.neck_conv; f.neck_pk = op.neck_pk;
                        f.out_buf = op.out_buf; f.out_ld = op.out_ld;
                        f.N = op.N;
                        snprintf(f.name, sizeof(f.name), "chain32_16_11blocks_neck_%dx%d", f.C, f.N);
                        f.flops += op.flops;
                        f.bytes = 4.0 * (1024.0 * f.C + 256.0 * f.N);
                    } else {
                        ops.push_back(op);
                    }
                    track(o);
                    pool.release(cur.buf);
                    cur = o;
                    bi += 8;
                    break;      // the neck is consumed: continue with the head
                }
            }
        }
        // ---- the whole 32 x 32 stage (three blocks + the stride-2 block down to 16 x 16) as one chain kernel (fp32, search branch)
        if (h->fuse && h->chain32 && mode != 3 && !h->math && !small && with_head && b.kind == FEARW_IR && cur.H == 32 && cur.W == 32 && cur.off == 0 &&
            bi + 3 < h->blocks.size()) {
            bool match = cur.C == kChain32XS[0].cin;
            for (int j = 0; j < 4 && match; ++j) {
                const FearwBlock& bj = h->blocks[bi + j];
                if (bj.kind != FEARW_IR || bj.conv[0] < 0) { match = false; break; }
                const Conv& e = h->convs[bj.conv[0]];
                const Conv& d = h->convs[bj.conv[1]];
                const Conv& p = h->convs[bj.conv[2]];
                const Chain32Shape& cs = kChain32XS[j];
                match = e.cin_g == cs.cin && d.cout == cs.cexp && p.cout == cs.cout && d.k == cs.ks && d.stride == cs.stride &&
                        (int)bj.residual == cs.res && e.has_bias && p.has_bias && e.relu && d.relu && !p.relu;
            }
            if (match) {
                Op op{};
                op.type = OP_CHAIN32;
                bool ok = true;
                double fl = 0;
                for (int j = 0; j < 4 && ok; ++j) {
                    const FearwBlock& bj = h->blocks[bi + j];
                    ok = pack_fused16(h, bj.conv[0], bj.conv[1], bj.conv[2], &op.chain_pk[j]) == FEAR_OK;
                    op.chain_cp[j] = bj.conv[2];
                    const Chain32Shape& cs = kChain32XS[j];
                    const double po = 1024.0 / (cs.stride * cs.stride);
                    fl += 2.0 * (1024.0 * cs.cin * cs.cexp + po * cs.ks * cs.ks * cs.cexp + po * cs.cexp * cs.cout);
                }
                if (ok) {
                    op.in_buf = cur.buf; op.in_ld = cur.ld; op.in_off = 0;
                    op.H = 32; op.W = 32; op.Ho = 16; op.Wo = 16; op.C = kChain32XS[0].cin; op.N = kChain32XS[3].cout;
                    T o;
                    o.buf = pool.acquire(); o.ld = op.N; o.off = 0; o.C = op.N; o.H = 16; o.W = 16;
                    op.out_buf = o.buf; op.out_ld = o.ld;
                    snprintf(op.name, sizeof(op.name), "chain32_4blocks_%dx%d_hw32", op.C, op.N);
                    op.flops = fl;
                    op.bytes = 4.0 * (1024.0 * op.C + 256.0 * op.N);
                    ops.push_back(op);
                    track(o);
                    pool.release(cur.buf);
                    cur = o;
                    bi += 3;        // four blocks consumed (the loop adds the fourth)
                    continue;
                }
            }
        }
        // ---- two consecutive e1 blocks (no expansion, 24 channels, residual) as one launch: the map between them stays in LDS
        //      (search branch only: `get_features` keeps one set of kernels for every batch size — its maps are bit-identical
        //      whatever the batching, tests/test_gpu_parity.py::test_empty_ragged_and_chunked_batches)
        if (h->fuse && h->e1_pair && !h->math && !small && with_head && b.kind == FEARW_IR && bi + 1 < h->blocks.size() && cur.C == E1PairGeom::C &&
            cur.ld == E1PairGeom::C && cur.off == 0 && cur.H == cur.W && cur.H % E1PairGeom::T == 0) {
            auto is_e1 = [&](const FearwBlock& q) {
                if (q.kind != FEARW_IR || q.conv[0] >= 0 || !q.residual) return false;
                const Conv& d = h->convs[q.conv[1]];
                const Conv& p = h->convs[q.conv[2]];
                return d.k == 3 && d.stride == 1 && d.cout == E1PairGeom::C && p.cin_g == E1PairGeom::C && p.cout == E1PairGeom::C &&
                       d.relu && !p.relu && p.has_bias;
            };
            if (is_e1(b) && is_e1(h->blocks[bi + 1])) {
                std::vector<float> pk;
                for (int j = 0; j < 2; ++j) {
                    const Conv& d = h->convs[h->blocks[bi + j].conv[1]];
                    const Conv& p = h->convs[h->blocks[bi + j].conv[2]];
                    e1pair_pack_block(pk, d.w.data(), d.has_bias ? d.b.data() : nullptr, p.w.data(), p.b.data());
                }
                Op op{};
                op.type = OP_E1PAIR;
                if (upload(h, pk, &op.d_packed) == FEAR_OK) {
                    op.in_buf = cur.buf; op.in_ld = cur.ld; op.in_off = 0;
                    op.H = cur.H; op.W = cur.W; op.Ho = cur.H; op.Wo = cur.W; op.C = cur.C; op.N = cur.C;
                    T o;
                    o.buf = pool.acquire(); o.ld = cur.C; o.off = 0; o.C = cur.C; o.H = cur.H; o.W = cur.W;
                    op.out_buf = o.buf; op.out_ld = o.ld;
                    snprintf(op.name, sizeof(op.name), "e1pair_%dx%d_k3_hw%d", cur.C, cur.C, cur.H);
                    op.flops = 2.0 * 2.0 * cur.H * cur.W * (9.0 * cur.C + (double)cur.C * cur.C);
                    op.bytes = 4.0 * 2.0 * cur.H * cur.W * cur.C;
                    ops.push_back(op);
                    track(o);
                    pool.release(cur.buf);
                    cur = o;
                    ++bi;           // two blocks consumed (the loop adds the other one)
                    continue;
                }
            }
        }
        if (b.kind == FEARW_IR) {
            T x = cur, e = cur, d, o;
            if (add_fused16(b.conv[0], b.conv[1], b.conv[2], x, o, b.residual ? &x : nullptr, 1, 0, 0, "ir16") ||
                add_fused_tile(b.conv[0], b.conv[1], b.conv[2], x, o, b.residual ? &x : nullptr)) {
                pool.release(x.buf);
                cur = o;
                continue;
            }
            if (b.conv[0] >= 0) {
                add_pw(b.conv[0], x, e, nullptr, 1, 0, 0, -1);
            }
            add_dw(b.conv[1], e, d, 1);
            if (b.conv[0] >= 0) pool.release(e.buf);
            add_pw(b.conv[2], d, o, b.residual ? &x : nullptr, 0, 0, 0, -1);
            pool.release(d.buf);
            pool.release(x.buf);
            cur = o;
        } else if (b.kind == FEARW_NECK) {
            T o;
            if (!with_head) {
                // write the features straight to the caller's NCHW tensor
                const Conv& c = h->convs[b.conv[0]];
                Op op{};
                op.type = OP_PW; op.conv = b.conv[0];
                op.in_buf = cur.buf; op.in_ld = cur.ld; op.in_off = cur.off;
                op.H = cur.H; op.W = cur.W; op.Ho = cur.H; op.Wo = cur.W; op.C = c.cin_g; op.N = c.cout;
                op.out_external = 1;
                set_name(op, "neck_%dx%d_hw%d", c.cin_g, c.cout, cur.H);
                op.flops = 2.0 * cur.H * cur.W * c.cin_g * c.cout;
                op.bytes = 4.0 * cur.H * cur.W * (c.cin_g + c.cout);
                ops.push_back(op);
                pool.release(cur.buf);
                cur = T{};
            } else {
                add_pw(b.conv[0], cur, o, nullptr, 0, 0, 0, -1);
                snprintf(ops.back().name, sizeof(ops.back().name), "neck_%dx%d_hw%d", ops.back().C, ops.back().N, cur.H);
                pool.release(cur.buf);
                cur = o;
            }
            ++bi;
            break;
        } else {
            return FEAR_ERR_FORMAT;
        }
    }
    // ---- bf16 activation storage for the HBM-bound front of the trunk (FEAR_OPT_MATH = 2, FEAR_OPT_BF16_STORE): a tile op writes
    //      bf16 when its output map is at least 64 x 64 and its consumer has a storage variant that reads it (kTileBf16)
    if (h->math == 2 && h->bf16_store && with_head && hw == 256 && !small && h->fuse) {
        auto reads_bf16 = [&](const Op& op) {      // some variant of this op takes a bf16 input (and residual, if it has one)
            if (op.type != OP_IRTILE || op.stem || op.splitk) return false;
            const int need = IO_X_BF16 | (op.res_buf >= 0 ? IO_R_BF16 : 0);
            return find_tile_bf16(0, op.fused_id, need) || find_tile_bf16(0, op.fused_id, need | IO_Y_BF16);
        };
        // writes[i]: op i stores its output as bf16.  Start from every candidate, then clear flags until each op has a storage
        // variant for the (input, residual, output) combination it ends up with: a consumer may exist only as "bf16 in AND out"
        // (kTileBf16 is a list of measured instantiations, not a full cross product), and whether ITS output goes out as bf16
        // depends on the op behind it.  A combination without a variant is never an error of the plan — this is an optional
        // optimisation — the producer in front of it just keeps fp32.
        std::vector<char> writes(ops.size(), 0);
        for (size_t i = 0; i + 1 < ops.size(); ++i)
            writes[i] = ops[i].type == OP_IRTILE && !ops[i].splitk && ops[i].Ho >= 64 && reads_bf16(ops[i + 1]);
        auto bits = [&](size_t i) {
            const Op& op = ops[i];
            const bool in_bf = i > 0 && writes[i - 1];
            return (in_bf ? (IO_X_BF16 | (op.res_buf >= 0 ? IO_R_BF16 : 0)) : 0) | (writes[i] ? IO_Y_BF16 : 0);
        };
        for (bool changed = true; changed;) {
            changed = false;
            for (size_t i = 0; i < ops.size(); ++i) {
                const Op& op = ops[i];
                if (op.type != OP_IRTILE || op.splitk) {
                    if (i > 0 && writes[i - 1]) { writes[i - 1] = 0; changed = true; }      // its input must be fp32
                    continue;
                }
                const int io = bits(i);
                if (!io || find_tile_bf16(op.stem, op.fused_id, io)) continue;
                if (writes[i]) writes[i] = 0;                  // first try this op with an fp32 output ...
                else if (i > 0 && writes[i - 1]) writes[i - 1] = 0;   // ... then hand it an fp32 input
                changed = true;
            }
        }
        for (size_t i = 0; i < ops.size(); ++i)
            if (ops[i].type == OP_IRTILE && !ops[i].splitk) ops[i].io_bf16 = bits(i);
    }
    // ---- head (BoxTower.forward, model/blocks.py:174-194)
    if (with_head) {
        const FearwBlock* role[9] = {nullptr};
        std::vector<const FearwBlock*> bbox_tower, cls_tower;
        for (; bi < h->blocks.size(); ++bi) {
            const FearwBlock& b = h->blocks[bi];
            if (b.kind != FEARW_SEP || b.role < 1 || b.role > 8) return FEAR_ERR_FORMAT;
            if (b.role == FEARW_BBOX_TOWER) bbox_tower.push_back(&b);
            else if (b.role == FEARW_CLS_TOWER) cls_tower.push_back(&b);
            else role[b.role] = &b;
        }
        for (int r : {FEARW_CLS_ENCODE, FEARW_REG_ENCODE, FEARW_CLS_CORR, FEARW_REG_CORR, FEARW_BBOX_PRED, FEARW_CLS_PRED})
            if (!role[r]) return FEAR_ERR_FORMAT;
        const T feat = cur;
        const int S = feat.H;               // 16 for a 256 search crop
        const int tz = 64;                  // template positions: (128/16)^2, matches fear_track's tmpl shape
        // one branch: encode -> correlation/concat -> corr sep -> towers -> pred
        auto branch = [&](const FearwBlock* enc, const FearwBlock* corr, const std::vector<const FearwBlock*>& tower,
                          const FearwBlock* pred, bool is_cls) -> int {
            const Conv& enc_pw = h->convs[enc->conv[1]];
            const Conv& corr_dw = h->convs[corr->conv[0]];
            if (corr_dw.cout != enc_pw.cout + tz) return FEAR_ERR_FORMAT;
            T d, cat;
            // encode pointwise writes channels [0, C) of the concat buffer; correlation fills [C, C+64)
            bool corr_done = false;
            if (!add_fused16(-1, enc->conv[0], enc->conv[1], feat, cat, nullptr, 0, 1, corr_dw.cout, "sep16")) {
                add_dw(enc->conv[0], feat, d, 0);
                add_pw(enc->conv[1], d, cat, nullptr, 1, corr_dw.cout, 0, -1);
                pool.release(d.buf);
            } else if (!h->math && !small && enc_pw.cout == kCorrC && h->convs[enc->conv[0]].cout == kCorrC &&
                       h->convs[enc->conv[0]].k == 3 && tz == kCorrTz) {
                // the correlation rides in the encode kernel's epilogue: its output fragments are the B operand as they stand
                Op& eop = ops.back();
                eop.corr_fused = 1;
                eop.tmpl_cls = is_cls ? 1 : 0;
                set_name(eop, is_cls ? "sep16_corr_cls_%dx%d_k%d" : "sep16_corr_reg_%dx%d_k%d", enc_pw.cout, tz, 3);
                eop.flops += 2.0 * S * S * enc_pw.cout * tz;
                eop.bytes += 4.0 * (S * S * tz + (double)enc_pw.cout * tz);
                corr_done = true;
            }
            if (!corr_done) {
                Op op{};
                op.type = OP_CORR;
                op.in_buf = cat.buf; op.in_ld = cat.ld; op.in_off = 0;
                op.out_buf = cat.buf; op.out_ld = cat.ld; op.out_off = enc_pw.cout;
                op.H = S; op.W = S; op.Ho = S; op.Wo = S; op.C = enc_pw.cout; op.N = tz;
                op.tmpl_cls = is_cls ? 1 : 0;
                set_name(op, is_cls ? "corr_cls_%dx%d_hw%d" : "corr_reg_%dx%d_hw%d", op.C, tz, S);
                op.flops = 2.0 * S * S * op.C * tz;
                op.bytes = 4.0 * (S * S * (op.C + tz) + (double)op.C * tz);
                ops.push_back(op);
            }
            T catv = cat;
            catv.C = corr_dw.cout;
            T x;
            if (add_fused16(-1, corr->conv[0], corr->conv[1], catv, x, nullptr, 0, 1, 0, "sep16")) {
                pool.release(cat.buf);
            } else {
                add_dw(corr->conv[0], catv, d, 0);
                pool.release(cat.buf);
                add_pw(corr->conv[1], d, x, nullptr, 1, 0, 0, -1);
                pool.release(d.buf);
            }
            for (const FearwBlock* tb : tower) {
                T y;
                if (add_fused16(-1, tb->conv[0], tb->conv[1], x, y, nullptr, 0, 1, 0, "sep16")) {
                    pool.release(x.buf);
                } else {
                    add_dw(tb->conv[0], x, d, 0);
                    pool.release(x.buf);
                    add_pw(tb->conv[1], d, y, nullptr, 1, 0, 0, -1);
                    pool.release(d.buf);
                }
                x = y;
            }
            if (h->convs[pred->conv[1]].cout != (is_cls ? 1 : 4)) return FEAR_ERR_FORMAT;
            // the prediction head rides in the epilogue of the last tower SepConv (its output never reaches HBM)
            if (h->fuse && !h->math && !small && !ops.empty() && ops.back().type == OP_IR16 && ops.back().out_buf == x.buf &&
                ops.back().C == 256 && ops.back().N == 256 && !ops.back().corr_fused && ops.back().res_buf < 0 &&
                h->convs[ops.back().conv_d].k == 3 && h->convs[pred->conv[0]].k == 3 && h->convs[pred->conv[0]].cout == 256 &&
                h->convs[pred->conv[0]].stride == 1 && h->convs[pred->conv[1]].has_bias && !h->convs[pred->conv[0]].relu) {
                Op& top = ops.back();
                if (pack_fused16(h, -1, pred->conv[0], pred->conv[1], &top.pred_packed) == FEAR_OK) {
                    const Conv& pc = h->convs[pred->conv[1]];
                    top.pred_fused = 1;
                    top.pred_conv_p = pred->conv[1];
                    top.pred_cout = pc.cout; top.act = pred->act; top.out_external = is_cls ? 3 : 2;
                    set_name(top, is_cls ? "sep16_cls_pred_%dx%dx%d" : "sep16_bbox_pred_%dx%dx%d", 256, 256, pc.cout);
                    top.flops += 2.0 * 256 * (256.0 * 9 + 256.0 * pc.cout);
                    top.bytes = 4.0 * 256 * (256 + pc.cout);
                    pool.release(x.buf);
                    return FEAR_OK;
                }
            }
            if (add_fused_pred(pred->conv[0], pred->conv[1], x, pred->act, is_cls ? 3 : 2)) {
                pool.release(x.buf);
                return FEAR_OK;
            }
            add_dw(pred->conv[0], x, d, 0);
            pool.release(x.buf);
            {
                const Conv& c = h->convs[pred->conv[1]];
                if (c.cout != (is_cls ? 1 : 4)) return FEAR_ERR_FORMAT;
                Op op{};
                op.type = OP_PW_SMALL; op.conv = pred->conv[1];
                op.in_buf = d.buf; op.in_ld = d.ld;
                op.H = S; op.W = S; op.Ho = S; op.Wo = S; op.C = c.cin_g; op.N = c.cout;
                op.act = pred->act;
                op.out_external = is_cls ? 3 : 2;
                set_name(op, is_cls ? "cls_pred_%dx%d_hw%d" : "bbox_pred_%dx%d_hw%d", c.cin_g, c.cout, S);
                op.flops = 2.0 * S * S * c.cin_g * c.cout;
                op.bytes = 4.0 * S * S * (c.cin_g + c.cout);
                ops.push_back(op);
            }
            pool.release(d.buf);
            return FEAR_OK;
        };
        // Small batches (one workgroup per crop leaves most CUs idle): the two branches are independent, so the bbox branch is
        // planned on buffers of its own and run_plan puts it on a second stream next to the cls branch.
        // The throughput plan does the same when FEAR_OPT_DUAL_HEAD is on (off by default): a sep16 workgroup needs 77 KB of LDS and
        // 116 VGPRs, so one workgroup of each branch fits on a CU at the same time — measured: no gain, the kernels are ALU-bound.
        // ---- the throughput plan in fp32: both branches, all eight SepConvs + correlations + prediction heads in ONE launch
        auto head_chain = [&]() -> bool {
            if (!h->head_chain || !h->fuse || h->math == 1 || small || S != 16 || feat.C != HeadChainG::C || bbox_tower.size() != 2 ||
                cls_tower.size() != 2)
                return false;
            // a later shape check or upload may still fail (and the plan then falls back to the sep16 launches): whatever this
            // attempt uploaded is freed again — up to twenty packed weight buffers per branch would otherwise stay booked in
            // plan_allocs until the plans are dropped, once per plan rebuild
            const size_t allocs0 = h->plan_allocs.size();
            auto undo = [&]() {
                while (h->plan_allocs.size() > allocs0) { hipFree(h->plan_allocs.back()); h->plan_allocs.pop_back(); }
                return false;
            };
            Op op{};
            op.type = OP_HEADCHAIN;
            op.math = h->math;          // 0: headchain_kernel (exact fp32) | 2: headchain_b_kernel (bf16 matrix pipe)
            double fl = 0;
            for (int br = 0; br < 2; ++br) {
                const bool is_cls = br == 0;
                const FearwBlock* seq[4] = {role[is_cls ? FEARW_CLS_ENCODE : FEARW_REG_ENCODE], role[is_cls ? FEARW_CLS_CORR : FEARW_REG_CORR],
                                            (is_cls ? cls_tower : bbox_tower)[0], (is_cls ? cls_tower : bbox_tower)[1]};
                const FearwBlock* pred = role[is_cls ? FEARW_CLS_PRED : FEARW_BBOX_PRED];
                std::vector<float> sep[4];
                for (int l = 0; l < 4; ++l) {
                    const Conv& d = h->convs[seq[l]->conv[0]];
                    const Conv& pw = h->convs[seq[l]->conv[1]];
                    const int cin = l == 1 ? HeadChainG::CC : HeadChainG::C;
                    // SepConv + BN + ReLU: depthwise 3x3 s1 without activation, pointwise to 256 with bias and ReLU
                    if (!d.is_dw() || !pw.is_pw() || d.k != 3 || d.stride != 1 || d.cout != cin || d.relu || pw.cin_g != cin ||
                        pw.cout != HeadChainG::C || !pw.has_bias || !pw.relu)
                        return undo();
                    sep[l] = pack_fused16_host(h, -1, seq[l]->conv[0], seq[l]->conv[1]);
                    fl += 2.0 * 256 * ((double)cin * 9 + (double)cin * HeadChainG::C);
                }
                const Conv& pd = h->convs[pred->conv[0]];
                const Conv& pp = h->convs[pred->conv[1]];
                if (!pd.is_dw() || !pp.is_pw() || pd.k != 3 || pd.stride != 1 || pd.cout != HeadChainG::C || pd.relu ||
                    pp.cin_g != HeadChainG::C || pp.cout != (is_cls ? 1 : 4) || !pp.has_bias)
                    return undo();
                if (h->math == 2) {
                    // bf16 fragments in the kernel's k order, fp32 bias / taps; prediction head: taps then 8 fragments of its 1x1
                    for (int l = 0; l < 4; ++l) {
                        const Conv& d = h->convs[seq[l]->conv[0]];
                        const Conv& pw = h->convs[seq[l]->conv[1]];
                        const int cin = l == 1 ? HeadChainG::CC : HeadChainG::C;
                        if (upload(h, headchain_b_pack(pw.w.data(), cin, HeadChainG::C, pw.b.data()), &op.hc_w[br][l]) != FEAR_OK ||
                            upload(h, headchain_b_taps(d.w.data(), d.has_bias ? d.b.data() : nullptr, cin, 3), &op.hc_taps[br][l]) != FEAR_OK)
                            return undo();
                    }
                    std::vector<float> pwk = headchain_b_taps(pd.w.data(), pd.has_bias ? pd.b.data() : nullptr, HeadChainG::C, 3);
                    for (int P = 0; P < HeadChainG::C / 32; ++P) headchain_b_push_frag(pwk, pp.w.data(), HeadChainG::C, pp.cout, 0, P);
                    if (upload(h, pwk, &op.hc_pred[br]) != FEAR_OK) return undo();
                } else {
                    for (int l = 0; l < 4; ++l) {
                        const Conv& pw = h->convs[seq[l]->conv[1]];
                        if (upload(h, headchain_pack(sep[l].data(), l == 1 ? HeadChainG::CC : HeadChainG::C, pw.b.data(),
                                                     l < 3 ? sep[l + 1].data() : nullptr, 3), &op.hc_w[br][l]) != FEAR_OK)
                            return undo();
                    }
                    if (upload(h, headchain_pack_dw(sep[0].data(), 0, HeadChainG::C / 16, 3), &op.hc_wd0[br]) != FEAR_OK ||
                        upload(h, headchain_pack_dw(sep[1].data(), HeadChainG::C / 16, HeadChainG::TZ / 16, 3), &op.hc_wdc[br]) != FEAR_OK ||
                        pack_fused16(h, -1, pred->conv[0], pred->conv[1], &op.hc_pred[br]) != FEAR_OK)
                        return undo();
                }
                op.hc_pred_conv[br] = pred->conv[1];
                op.hc_pred_act[br] = pred->act;
                fl += 2.0 * 256 * ((double)HeadChainG::C * HeadChainG::TZ + (double)HeadChainG::C * 9 + (double)HeadChainG::C * pp.cout);
            }
            op.in_buf = feat.buf; op.in_ld = feat.ld; op.in_off = feat.off;
            op.H = S; op.W = S; op.Ho = S; op.Wo = S; op.C = feat.C; op.N = 5;
            op.relu_dw = 0; op.relu = 1;
            // scratch of the depthwise results, both branches of a crop side by side
            op.out_buf = pool.acquire();
            if ((size_t)2 * HeadChainG::D_FLOATS > max_elems) max_elems = (size_t)2 * HeadChainG::D_FLOATS;
            snprintf(op.name, sizeof(op.name), h->math == 2 ? "headchain_bf16_boxtower_%dx%d" : "headchain_boxtower_%dx%d", feat.C, tz);
            op.flops = fl;
            op.bytes = 4.0 * (S * S * feat.C + 2.0 * feat.C * tz + 5.0 * S * S);
            ops.push_back(op);
            pool.release(op.out_buf);
            return true;
        };
        if (head_chain()) {
            pool.release(feat.buf);
            plan->n_bufs = pool.count;
            plan->buf_floats_per_crop = (max_elems + 63) & ~(size_t)63;
            *out = plan.get();
            h->plans[key] = std::move(plan);
            return FEAR_OK;
        }
        const bool dual = small || (h->dual_head && h->fuse && !h->math);
        const size_t head_first = ops.size();
        pool.hold = dual;
        int st = branch(role[FEARW_CLS_ENCODE], role[FEARW_CLS_CORR], cls_tower, role[FEARW_CLS_PRED], true);
        if (st != FEAR_OK) return st;
        const size_t reg_first = ops.size();
        st = branch(role[FEARW_REG_ENCODE], role[FEARW_REG_CORR], bbox_tower, role[FEARW_BBOX_PRED], false);
        if (st != FEAR_OK) return st;
        pool.flush();
        if (dual) {
            for (size_t i = reg_first; i < ops.size(); ++i) ops[i].lane = 1;
            plan->head_first = (int)head_first;
        }
        pool.release(feat.buf);
    }
    if (small)
        for (Op& op : ops)
            if (op.type == OP_PW || op.type == OP_CORR) op.pw_split = 1;
    plan->n_bufs = pool.count;
    plan->buf_floats_per_crop = (max_elems + 63) & ~(size_t)63;
    *out = plan.get();
    h->plans[key] = std::move(plan);
    return FEAR_OK;
}

// crops a pass of this plan can hold: the small-batch plan never sees more than FEAR_OPT_SMALL_PASS crops
size_t pass_cap(const fear_handle* h, const Plan& p) {
    return p.small && h->small_pass > 0 && h->small_pass < h->max_batch ? (size_t)h->small_pass : (size_t)h->max_batch;
}

int ensure_workspace(fear_handle* h, const Plan& p) {
    const size_t need = (size_t)p.n_bufs * p.buf_floats_per_crop * pass_cap(h, p);
    if (need <= h->workspace_floats) return FEAR_OK;
    if (h->workspace) {
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipFree(h->workspace));
        h->workspace = nullptr;
        h->workspace_floats = 0;
    }
    if (hipMalloc(&h->workspace, need * sizeof(float)) != hipSuccess) return FEAR_ERR_ALLOC;
    h->workspace_floats = need;
    return FEAR_OK;
}

template <int MT, bool WKN>
void launch_pw_nt(int nt, dim3 grid, hipStream_t s, const PwArgs& a) {
    switch (nt) {
        case 1: hipLaunchKernelGGL((pw_mfma_kernel<MT, 1, WKN, FEAR_PW_KU(1)>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((pw_mfma_kernel<MT, 2, WKN, FEAR_PW_KU(2)>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((pw_mfma_kernel<MT, 3, WKN, FEAR_PW_KU(3)>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pw_mfma_kernel<MT, 4, WKN, FEAR_PW_KU(4)>), grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((pw_mfma_kernel<MT, 6, WKN, FEAR_PW_KU(6)>), grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((pw_mfma_kernel<MT, 7, WKN, FEAR_PW_KU(7)>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((pw_mfma_kernel<MT, 8, WKN, FEAR_PW_KU(8)>), grid, dim3(256), 0, s, a); break;
    }
}

// FEAR_OPT_MATH = 1 / 2: the same GEMM on the matrix pipe (pw_h_kernel); 4, 2 or 1 channel tiles per pass
template <bool WKN>
void launch_pw_h(int math, int n_tiles, dim3 grid, hipStream_t s, const PwArgs& a) {
    const int nt = n_tiles % 4 == 0 ? 4 : n_tiles % 2 == 0 ? 2 : 1;
    if (math == 2) {
        if (nt == 4) hipLaunchKernelGGL((pw_h_kernel<2, 4, WKN, 2>), grid, dim3(256), 0, s, a);
        else if (nt == 2) hipLaunchKernelGGL((pw_h_kernel<2, 2, WKN, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pw_h_kernel<2, 1, WKN, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (nt == 4) hipLaunchKernelGGL((pw_h_kernel<2, 4, WKN, 1>), grid, dim3(256), 0, s, a);
        else if (nt == 2) hipLaunchKernelGGL((pw_h_kernel<2, 2, WKN, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pw_h_kernel<2, 1, WKN, 1>), grid, dim3(256), 0, s, a);
    }
}

// channel tiles per pass: the largest of {8,7,6,4,3,2,1} that divides the tile count
int pick_nt(int n_tiles) {
    for (int nt : {8, 7, 6, 4, 3, 2, 1})
        if (n_tiles % nt == 0) return nt;
    return 1;
}

template <int KS, int S>
void launch_dw(dim3 grid, hipStream_t s, const DwArgs& a) {
    hipLaunchKernelGGL((dw_conv_kernel<KS, S, 4>), grid, dim3(256), 0, s, a);
}

struct Ext {
    const float* img;      // NCHW input
    const float* tmpl;     // template features
    const float* tmpl_cls;
    float* feat_out;
    float* bbox_out;
    float* cls_out;
    long bbox_stride = 4 * 256, cls_stride = 256;   // floats between consecutive crops' maps (fear_track_packed: 5 * 256 both)
};

// One wavefront that does nothing for `ticks` x 10 ns (wall_clock64 runs at 100 MHz) and at most ~1 ms: put in front of the
// head's second branch (FEAR_OPT_HEAD_STAGGER) so that the two branches' kernels — co-resident, one workgroup of each per CU —
// run half a kernel out of phase: one branch's prologue / output burst then falls into the other's MFMA stretch.
__global__ void delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < 100000 && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}

// crop0 / sub: the second half of a split pass (FEAR_OPT_SPLIT_STREAMS) — its crops use the workspace behind the first half's
// (every pool buffer is sized for a full pass at the largest per-crop footprint, so crop c0's share starts c0 footprints in), and the
// caller (track_impl) does the stream bookkeeping around both halves
int run_plan(fear_handle* h, Plan& p, int n, const Ext& ext, hipStream_t s_main, size_t crop0 = 0, bool sub = false) {
    if (!h->fused_attr_set) {
        for (const Fused16& f : kFused16) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
            if (f.kernel_splitk)
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel_splitk),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
            if (f.kernel_tiny)
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel_tiny),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_tiny));
        }
        for (const Fused16& f : kFused16H)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const FusedTile& f : kFusedTileV4)
            if (f.kernel)
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const FusedTile& f : kFusedTile)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const FusedTile& f : kFusedTileSmall)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const FusedTile& f : kFusedTileTiny)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const TileKSplit& f : kTileKSplit)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const FusedTile& f : kFusedTileH)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const FusedTile& f : kFusedTileB)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        for (const Fused16& f : kFused16B)
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(f.kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f.lds_bytes));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kSep16PredKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSep16PredLds));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kSep16CorrKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSep16CorrLds));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kChainXSKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kChainXSLds));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kChain32XSKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C32Geom::LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kChain32_16XSKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kChain32_16Lds));
        for (const TileBf16& tb : kTileBf16) {
            const int lds = tb.stem ? kStemTile.lds_bytes : (tb.id == 1 || tb.id == 3) ? kFusedTileB[tb.id].lds_bytes : kFusedTile[tb.id].lds_bytes;
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(tb.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        }
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(e1pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, E1PairGeom::LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kHeadChainKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, HeadChainG::LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kHeadChainBKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, HeadChainBG::LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kStemTile.kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kStemTile.lds_bytes));
        h->fused_attr_set = true;
    }
    // one workspace per handle: a call on another stream than the previous one must not overtake it
    // (the event was recorded on the previous call's stream at the END of that call, below: the old stream handle is only
    // compared here, never used — the caller may have destroyed that stream since)
    if (!h->stream_switch) HIP_TRY(h, hipEventCreateWithFlags(&h->stream_switch, hipEventDisableTiming));
    if (!sub) {
        if (h->last_stream_valid && h->last_stream != s_main) HIP_TRY(h, hipStreamWaitEvent(s_main, h->stream_switch, 0));
        h->last_stream = s_main;
        h->last_stream_valid = true;
    }
    const size_t slab = p.buf_floats_per_crop * pass_cap(h, p);
    auto buf = [&](int id) -> float* { return h->workspace + (size_t)id * slab + crop0 * p.buf_floats_per_crop; };
    // head branches on two streams (plans built for small passes only; per-op profiling keeps everything on one stream)
    // (bracketing ONE op with events keeps the two streams — the events go to the stream the op runs on; the all-ops table of
    // FEAR_OPT_PROFILE_OP = -1 serialises the plan so that per-op times do not overlap)
    const bool dual = p.head_first >= 0 && !(h->profile && h->profile_op < 0);
    if (dual && !h->branch_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->branch_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->branch_fork, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->branch_join, hipEventDisableTiming));
    }
    bool forked = false;
    int op_index = -1;
    for (Op& op : p.ops) {
        ++op_index;
        if (dual && op_index == p.head_first) {
            HIP_TRY(h, hipEventRecord(h->branch_fork, s_main));            // the trunk's output is ready after this point
            HIP_TRY(h, hipStreamWaitEvent(h->branch_stream, h->branch_fork, 0));
            if (h->head_stagger_us > 0)
                hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, h->branch_stream, (long long)h->head_stagger_us * 100);
            forked = true;
        }
        const hipStream_t s = (dual && op.lane == 1) ? h->branch_stream : s_main;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        // a selected op also selects every op with the same name (= same kernel symbol and shape)
        const bool prof = h->profile && (h->profile_op < 0 || h->profile_op == op_index ||
                                         (h->profile_op < (int)p.ops.size() && !strcmp(p.ops[h->profile_op].name, op.name)));
        if (prof) {
            for (hipEvent_t* e : {&e0, &e1}) {
                if (!h->event_pool.empty()) { *e = h->event_pool.back(); h->event_pool.pop_back(); }
                else HIP_TRY(h, hipEventCreate(e));
            }
            HIP_TRY(h, hipEventRecord(e0, s));
        }
        const Conv* c = op.conv >= 0 ? &h->convs[op.conv] : nullptr;
        switch (op.type) {
            case OP_STEM: {
                if (h->fuse && op.Wo % 16 == 0) {
                    StemMfmaArgs a{ext.img, c->d_w2, c->d_b, buf(op.out_buf), n, op.H, op.W, op.Ho, op.Wo};
                    const long rows = (long)n * op.Ho;
                    hipLaunchKernelGGL(stem_mfma_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, a);
                } else {
                    StemArgs a{ext.img, c->d_w, c->d_b, buf(op.out_buf), n, op.H, op.W, op.Ho, op.Wo};
                    const long total = (long)n * op.Ho * op.Wo;
                    hipLaunchKernelGGL(stem_conv_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
                }
                break;
            }
            case OP_PW: {
                PwArgs a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.W = c->d_w; a.bias = c->d_b;
                a.R = op.res_buf >= 0 ? buf(op.res_buf) : nullptr; a.ldr = op.res_ld;
                a.M = n * op.H * op.W; a.K = op.C; a.N = op.N; a.relu = op.relu;
                if (op.out_external == 1) { a.Y = ext.feat_out; a.ldy = op.N; a.nchw_hw = op.H * op.W; }
                else { a.Y = buf(op.out_buf) + op.out_off; a.ldy = op.out_ld; a.nchw_hw = 0; }
                const int n_tiles = (op.N + 15) / 16;
                int nt = pick_nt(n_tiles);
                const int rows_per_block = 4 * 2 * 16;
                dim3 grid((a.M + rows_per_block - 1) / rows_per_block);
                if (op.pw_split && n_tiles % 2 == 0) {
                    grid.y = n_tiles / 2;
                    hipLaunchKernelGGL((pw_mfma_kernel<2, 2, false, 8>), grid, dim3(256), 0, s, a);
                    break;
                }
                if (h->math && p.with_head) launch_pw_h<false>(h->math, n_tiles, grid, s, a);   // (the template branch's features stay fp32)
                else launch_pw_nt<2, false>(nt, grid, s, a);
                break;
            }
            case OP_CORR: {
                PwArgs a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.W = (op.tmpl_cls && ext.tmpl_cls) ? ext.tmpl_cls : ext.tmpl;
                a.bias = nullptr; a.R = nullptr;
                a.Y = buf(op.out_buf) + op.out_off; a.ldy = op.out_ld;
                a.M = n * op.H * op.W; a.K = op.C; a.N = op.N; a.relu = 0; a.nchw_hw = 0;
                a.rows_per_crop = op.H * op.W; a.w_crop_stride = (long)op.C * op.N;
                const int rows_per_block = 4 * 2 * 16;
                dim3 grid((a.M + rows_per_block - 1) / rows_per_block);
                int nt = pick_nt(op.N / 16);
                if (op.pw_split) {
                    // 16 rows per wave and the whole K = 256 in one batch of loads: 16 workgroups per crop, one memory round trip
                    grid.x = (a.M + 63) / 64;
                    grid.y = op.N / 16;
                    if (a.K <= 256) hipLaunchKernelGGL((pw_mfma_kernel<1, 1, true, 16>), grid, dim3(256), 0, s, a);
                    else hipLaunchKernelGGL((pw_mfma_kernel<1, 1, true, 8>), grid, dim3(256), 0, s, a);
                    break;
                }
                if (h->math) launch_pw_h<true>(h->math, op.N / 16, grid, s, a);
                else launch_pw_nt<2, true>(nt, grid, s, a);
                break;
            }
            case OP_DW: {
                DwArgs a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.Wt = c->d_w; a.bias = c->d_b;
                a.Y = buf(op.out_buf); a.ldy = op.out_ld;
                a.B = n; a.H = op.H; a.W = op.W; a.C = op.C; a.Ho = op.Ho; a.Wo = op.Wo; a.relu = op.relu;
                const long strips = (op.Ho + 3) / 4;
                const long total = (long)n * strips * op.Wo * (op.C / 4);
                dim3 grid((total + 255) / 256);
                if (c->k == 3 && c->stride == 1) launch_dw<3, 1>(grid, s, a);
                else if (c->k == 3 && c->stride == 2) launch_dw<3, 2>(grid, s, a);
                else if (c->k == 5 && c->stride == 1) launch_dw<5, 1>(grid, s, a);
                else launch_dw<5, 2>(grid, s, a);
                break;
            }
            case OP_IR16: {
                const Fused16& f = op.math == 2 ? kFused16B[op.fused_id] : op.math ? kFused16H[op.fused_id] : kFused16[op.fused_id];
                Ir2Args a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.Wpk = op.d_packed; a.bp = h->convs[op.conv_p].d_b;
                a.R = op.res_buf >= 0 ? buf(op.res_buf) : nullptr; a.ldr = op.res_ld;
                a.Y = op.out_buf >= 0 ? buf(op.out_buf) : nullptr; a.ldy = op.out_ld;
                a.relu_dw = op.relu_dw; a.relu_out = op.relu;
                if (op.pred_fused) {
                    a.pred_cout = op.pred_cout; a.pred_act = op.act;
                    a.P_Wpk = op.pred_packed; a.P_bp = h->convs[op.pred_conv_p].d_b;
                    a.P_Y = op.out_external == 3 ? ext.cls_out : ext.bbox_out;
                    a.pred_stride = op.out_external == 3 ? ext.cls_stride : ext.bbox_stride;
                } else if (op.pred_cout > 0) {
                    a.pred_cout = op.pred_cout; a.pred_act = op.act;
                    a.Y = op.out_external == 3 ? ext.cls_out : ext.bbox_out;
                    a.pred_stride = op.out_external == 3 ? ext.cls_stride : ext.bbox_stride;
                }
                if (op.nsplit) {
                    const Conv& dconv = h->convs[op.conv_d];
                    a.nsplit_wstride = (long)(dconv.cout / 16) * (256 + dconv.k * dconv.k * 16 + 16);
                    if (op.tiny) hipLaunchKernelGGL(f.kernel_tiny, dim3(n, op.nsplit, 16 / kTinyRows), dim3(64 * kTinyRows * kTinyKSplit), f.lds_tiny, s, a);
                    else hipLaunchKernelGGL(f.kernel, dim3(n, op.nsplit), dim3(512), f.lds_bytes, s, a);
                } else if (op.tiny) {
                    hipLaunchKernelGGL(f.kernel_tiny, dim3(n, 1, 16 / kTinyRows), dim3(64 * kTinyRows * kTinyKSplit), f.lds_tiny, s, a);
                } else if (op.splitk) {
                    // several workgroups per crop, each over its own chunk range -> partial projections -> reduce
                    Ir2Args pa = a;
                    const Conv& dconv = h->convs[op.conv_d];
                    pa.kc_count = dconv.cout / 16 / op.splitk;
                    pa.Y = buf(op.part_buf); pa.ldy = op.N;
                    pa.kc_part_stride = (long)n * 256 * op.N;
                    pa.R = nullptr;
                    hipLaunchKernelGGL(f.kernel_splitk, dim3(n, op.splitk), dim3(512), f.lds_bytes, s, pa);
                    SplitKReduceArgs ra{};
                    ra.P = pa.Y; ra.bias = a.bp; ra.R = a.R; ra.Y = a.Y; ra.part_stride = pa.kc_part_stride;
                    ra.W = op.splitk; ra.M = n * 256; ra.N = op.N; ra.ldp = op.N; ra.ldr = a.ldr; ra.ldy = a.ldy; ra.relu = a.relu_out;
                    const long total = (long)ra.M * (ra.N / 4);
                    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ra);
                } else if (op.corr_fused) {
                    a.Z = (op.tmpl_cls && ext.tmpl_cls) ? ext.tmpl_cls : ext.tmpl;
                    a.z_stride = (long)kCorrC * kCorrTz;
                    hipLaunchKernelGGL(kSep16CorrKernel, dim3(n), dim3(512), kSep16CorrLds, s, a);
                } else if (op.pred_fused) {
                    hipLaunchKernelGGL(kSep16PredKernel, dim3(n), dim3(512), kSep16PredLds, s, a);
                } else {
                    hipLaunchKernelGGL(f.kernel, dim3(n), dim3(512), f.lds_bytes, s, a);
                }
                break;
            }
            case OP_IRTILE: {
                const FusedTile& f = op.stem ? kStemTile : (op.math == 2 ? kFusedTileB[op.fused_id] : op.math ? kFusedTileH[op.fused_id] :
                                                            fp32_tile(op.small_tiles, op.fused_id));
                IrT2Args ta{};
                Ir2Args& a = ta.b;
                if (op.stem) { a.X = ext.img; a.ldx = 0; }
                else { a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld; }
                a.Wpk = op.d_packed; a.bp = h->convs[op.conv_p].d_b;
                a.R = op.res_buf >= 0 ? buf(op.res_buf) : nullptr; a.ldr = op.res_ld;
                a.Y = buf(op.out_buf); a.ldy = op.out_ld;
                a.relu_dw = op.relu_dw; a.relu_out = op.relu;
                ta.H = op.H; ta.W = op.W; ta.tiles_x = op.Wo / f.tw; ta.tiles_y = op.Ho / f.th;
                if (op.splitk) {
                    const TileKSplit* ks = find_tile_ksplit(op.fused_id);
                    SplitKReduceArgs ra{};
                    ra.bias = a.bp; ra.R = a.R; ra.Y = a.Y; ra.W = op.splitk; ra.M = n * op.Ho * op.Wo; ra.N = op.N; ra.ldp = op.N;
                    ra.ldr = a.ldr; ra.ldy = a.ldy; ra.relu = a.relu_out;
                    ra.part_stride = (long)ra.M * op.N;
                    a.Y = buf(op.part_buf); a.ldy = op.N; a.R = nullptr; a.kc_part_stride = ra.part_stride;
                    ra.P = a.Y;
                    hipLaunchKernelGGL(ks->kernel, dim3((unsigned)n * ta.tiles_x * ta.tiles_y, op.splitk), dim3(512), ks->lds_bytes, s, ta);
                    const long total = (long)ra.M * (ra.N / 4);
                    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ra);
                    break;
                }
                if (op.io_bf16) {      // same tile, storage variant (plan: bf16 front of the trunk)
                    const TileBf16* tb = find_tile_bf16(op.stem, op.fused_id, op.io_bf16);
                    hipLaunchKernelGGL(tb->kernel, dim3((unsigned)n * ta.tiles_x * ta.tiles_y), dim3(64 * f.nw), f.lds_bytes, s, ta);
                    break;
                }
                hipLaunchKernelGGL(f.kernel, dim3((unsigned)n * ta.tiles_x * ta.tiles_y), dim3(64 * f.nw), f.lds_bytes, s, ta);
                break;
            }
            case OP_E1PAIR: {
                E1PairArgs a{};
                a.X = buf(op.in_buf); a.Y = buf(op.out_buf); a.Wpk = op.d_packed;
                a.H = op.H; a.W = op.W; a.tiles_x = op.W / E1PairGeom::T; a.tiles_y = op.H / E1PairGeom::T;
                // consecutive tiles per workgroup (the next tile's loads run under the current tile's arithmetic), never fewer than two
                // workgroups for each of the 256 CUs; FEAR_E1PAIR_TPW_MAX = 1: one tile per workgroup
                const unsigned total = (unsigned)n * a.tiles_x * a.tiles_y;
                a.tpw = 1;
                for (int t = FEAR_E1PAIR_TPW_MAX; t > 1; t >>= 1)
                    if (total % t == 0 && total / t >= 512) { a.tpw = t; break; }
                hipLaunchKernelGGL(e1pair_kernel, dim3(total / a.tpw), dim3(512), E1PairGeom::LDS_BYTES, s, a);
                break;
            }
            case OP_CHAIN16: {
                Chain16Args a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.Y = buf(op.out_buf); a.ldy = op.out_ld;
                for (int j = 0; j < 7; ++j) { a.Wpk[j] = op.chain_pk[j]; a.bp[j] = h->convs[op.chain_cp[j]].d_b; }
                a.neck_pk = op.neck_pk; a.neck_b = h->convs[op.neck_conv].d_b;
                hipLaunchKernelGGL(kChainXSKernel, dim3(n), dim3(512), kChainXSLds, s, a);
                break;
            }
            case OP_CHAIN32: {
                Chain32Args a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.Y = buf(op.out_buf); a.ldy = op.out_ld;
                for (int j = 0; j < 4; ++j) { a.Wpk[j] = op.chain_pk[j]; a.bp[j] = h->convs[op.chain_cp[j]].d_b; }
                hipLaunchKernelGGL(kChain32XSKernel, dim3(n), dim3(512), C32Geom::LDS_BYTES, s, a);
                break;
            }
            case OP_CHAIN32_16: {
                Chain32Args a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld;
                a.Y = nullptr; a.ldy = 0;
                for (int j = 0; j < 4; ++j) { a.Wpk[j] = op.chain_pk[j]; a.bp[j] = h->convs[op.chain_cp[j]].d_b; }
                Chain16Args b{};
                b.X = nullptr; b.ldx = 0;
                b.Y = buf(op.out_buf); b.ldy = op.out_ld;
                for (int j = 0; j < 7; ++j) { b.Wpk[j] = op.chain16_pk[j]; b.bp[j] = h->convs[op.chain16_cp[j]].d_b; }
                b.neck_pk = op.neck_pk; b.neck_b = h->convs[op.neck_conv].d_b;
                hipLaunchKernelGGL(kChain32_16XSKernel, dim3(n), dim3(512), kChain32_16Lds, s, a, b);
                break;
            }
            case OP_HEADCHAIN: {
                if (op.math == 2) {
                    HeadChainBArgs a{};
                    a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld; a.n_crops = n;
                    a.relu_dw = op.relu_dw; a.relu_out = op.relu;
                    for (int br = 0; br < 2; ++br) {
                        HeadChainBBranch& b = a.br[br];
                        for (int l = 0; l < 4; ++l) { b.W[l] = op.hc_w[br][l]; b.Wd[l] = op.hc_taps[br][l]; }
                        b.Z = (br == 0 && ext.tmpl_cls) ? ext.tmpl_cls : ext.tmpl;
                        b.z_stride = (long)HeadChainG::C * HeadChainG::TZ;
                        b.P_W = op.hc_pred[br]; b.P_bp = h->convs[op.hc_pred_conv[br]].d_b;
                        b.P_Y = br == 0 ? ext.cls_out : ext.bbox_out;
                        b.pred_stride = br == 0 ? ext.cls_stride : ext.bbox_stride;
                        b.pred_cout = br == 0 ? 1 : 4; b.pred_act = op.hc_pred_act[br];
                    }
                    hipLaunchKernelGGL(kHeadChainBKernel, dim3(16u * ((unsigned)(n + 7) / 8)), dim3(512), HeadChainBG::LDS_BYTES, s, a);
                    break;
                }
                HeadChainArgs a{};
                a.X = buf(op.in_buf) + op.in_off; a.ldx = op.in_ld; a.n_crops = n;
                a.relu_dw = op.relu_dw; a.relu_out = op.relu;
                for (int br = 0; br < 2; ++br) {
                    HeadChainBranch& b = a.br[br];
                    for (int l = 0; l < 4; ++l) b.W[l] = op.hc_w[br][l];
                    b.Wd0 = op.hc_wd0[br]; b.WdC = op.hc_wdc[br];
                    b.Z = (br == 0 && ext.tmpl_cls) ? ext.tmpl_cls : ext.tmpl;
                    b.z_stride = (long)HeadChainG::C * HeadChainG::TZ;
                    b.P_Wpk = op.hc_pred[br]; b.P_bp = h->convs[op.hc_pred_conv[br]].d_b;
                    b.P_Y = br == 0 ? ext.cls_out : ext.bbox_out;
                    b.pred_stride = br == 0 ? ext.cls_stride : ext.bbox_stride;
                    b.pred_cout = br == 0 ? 1 : 4; b.pred_act = op.hc_pred_act[br];
                    b.D = buf(op.out_buf) + (size_t)br * n * HeadChainG::D_FLOATS;
                }
                hipLaunchKernelGGL(kHeadChainKernel, dim3(16u * ((unsigned)(n + 7) / 8)), dim3(512), HeadChainG::LDS_BYTES, s, a);
                break;
            }
            case OP_PW_SMALL: {
                PwSmallArgs a{};
                a.X = buf(op.in_buf); a.ldx = op.in_ld; a.W = c->d_w; a.bias = c->d_b;
                a.Y = op.out_external == 3 ? ext.cls_out : ext.bbox_out;
                a.M = n * op.H * op.W; a.K = op.C; a.N = op.N; a.hw = op.H * op.W; a.act = op.act;
                a.crop_stride = op.out_external == 3 ? ext.cls_stride : ext.bbox_stride;
                dim3 grid(((long)a.M * 16 + 255) / 256);
                if (op.N == 1) hipLaunchKernelGGL((pw_small_kernel<1>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((pw_small_kernel<4>), grid, dim3(256), 0, s, a);
                break;
            }
        }
        if (prof) {
            HIP_TRY(h, hipEventRecord(e1, s));
            op.events.emplace_back(e0, e1);
        }
    }
    if (forked) {                                    // the caller's stream continues once the bbox branch is done too
        HIP_TRY(h, hipEventRecord(h->branch_join, h->branch_stream));
        HIP_TRY(h, hipStreamWaitEvent(s_main, h->branch_join, 0));
    }
    if (!sub) HIP_TRY(h, hipEventRecord(h->stream_switch, s_main));      // what a later call on ANOTHER stream waits for (see above)
    HIP_TRY(h, hipGetLastError());
    return FEAR_OK;
}

int drain_events(fear_handle* h) {
    for (auto& kv : h->plans)
        for (Op& op : kv.second->ops) {
            for (auto& ev : op.events) {
                HIP_TRY(h, hipEventSynchronize(ev.second));
                float ms = 0.f;
                HIP_TRY(h, hipEventElapsedTime(&ms, ev.first, ev.second));
                op.prof_ms += ms;
                op.prof_n += 1;
                h->event_pool.push_back(ev.first);
                h->event_pool.push_back(ev.second);
            }
            op.events.clear();
        }
    return FEAR_OK;
}

// Forget every cached plan (an option that shapes the plans changed): finish outstanding work, fold and recycle the
// profiling events the ops still own, free the weights that were packed for those plans.
int drop_plans(fear_handle* h) {
    if (h->plans.empty() && h->plan_allocs.empty()) return FEAR_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const int st = drain_events(h);
    if (st != FEAR_OK) return st;
    h->plans.clear();
    for (float* p : h->plan_allocs) hipFree(p);
    h->plan_allocs.clear();
    return FEAR_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* fear_version(void) { return "feartracker_amd 0.1 (gfx950, fp32 MFMA)"; }

const char* fear_strerror(int status) {
    switch (status) {
        case FEAR_OK: return "ok";
        case FEAR_ERR_NULL: return "null handle or pointer";
        case FEAR_ERR_SHAPE: return "unsupported shape or option value";
        case FEAR_ERR_FORMAT: return "malformed or unsupported .fearw model";
        case FEAR_ERR_HIP: return "HIP runtime error";
        case FEAR_ERR_ALLOC: return "allocation failed";
        case FEAR_ERR_NOHEAD: return "model has no correlation head";
        default: return "unknown status";
    }
}

int fear_create(const void* fearw_blob, size_t nbytes, int device, fear_handle** out) {
    if (!fearw_blob || !out) return FEAR_ERR_NULL;
    *out = nullptr;
    std::unique_ptr<fear_handle> h(new (std::nothrow) fear_handle);
    if (!h) return FEAR_ERR_ALLOC;
    h->device = device;
    int st = parse_model(h.get(), static_cast<const uint8_t*>(fearw_blob), nbytes);
    if (st != FEAR_OK) return st;
    if (hipSetDevice(device) != hipSuccess) return FEAR_ERR_HIP;
    st = pack_weights(h.get());
    if (st != FEAR_OK) {
        for (float* p : h->weight_allocs) hipFree(p);
        return st;
    }
    *out = h.release();
    return FEAR_OK;
}

int fear_destroy(fear_handle* h) {
    if (!h) return FEAR_ERR_NULL;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    drain_events(h);
    for (hipEvent_t e : h->event_pool) hipEventDestroy(e);
    if (h->split_stream) hipStreamDestroy(h->split_stream);
    if (h->split_fork) hipEventDestroy(h->split_fork);
    if (h->split_join) hipEventDestroy(h->split_join);
    if (h->branch_stream) hipStreamDestroy(h->branch_stream);
    if (h->branch_fork) hipEventDestroy(h->branch_fork);
    if (h->branch_join) hipEventDestroy(h->branch_join);
    if (h->stream_switch) hipEventDestroy(h->stream_switch);
    for (float* p : h->weight_allocs) hipFree(p);
    for (float* p : h->plan_allocs) hipFree(p);
    if (h->workspace) hipFree(h->workspace);
    delete h;
    return FEAR_OK;
}

int fear_set_option(fear_handle* h, int option, int64_t value) {
    if (!h) return FEAR_ERR_NULL;
    switch (option) {
        case FEAR_OPT_MAX_BATCH:
            if (value < 1 || value > 65536) return FEAR_ERR_SHAPE;
            h->max_batch = (int)value;
            return FEAR_OK;
        case FEAR_OPT_PROFILE:
            h->profile = value ? 1 : 0;
            return FEAR_OK;
        case FEAR_OPT_PROFILE_OP:
            if (value < -1 || value > 4096) return FEAR_ERR_SHAPE;
            h->profile_op = (int)value;
            return FEAR_OK;
        case FEAR_OPT_FUSE:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->fuse != (int)value) { h->fuse = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_MATH:
            if (value < 0 || value > 2) return FEAR_ERR_SHAPE;
            if (h->math != (int)value) { h->math = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_CHAIN:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->chain != (int)value) { h->chain = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_SMALL_PASS:
            if (value < 0 || value > 65536) return FEAR_ERR_SHAPE;
            h->small_pass = (int)value;
            return FEAR_OK;
        case FEAR_OPT_PLAN_CROPS:
            if (value < 0 || value > 65536) return FEAR_ERR_SHAPE;
            h->plan_crops = (int)value;
            return FEAR_OK;
        case FEAR_OPT_DUAL_HEAD:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->dual_head != (int)value) { h->dual_head = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_TILE_V4:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->tile_v4 != (int)value) { h->tile_v4 = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_TINY_SEP:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->tiny_sep != (int)value) { h->tiny_sep = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_HEAD_STAGGER:
            if (value < 0 || value > 1000) return FEAR_ERR_SHAPE;
            h->head_stagger_us = (int)value;
            return FEAR_OK;
        case FEAR_OPT_HEAD_CHAIN:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->head_chain != (int)value) { h->head_chain = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_E1_PAIR:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->e1_pair != (int)value) { h->e1_pair = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_CHAIN32:
            if (value < 0 || value > 2) return FEAR_ERR_SHAPE;
            if (h->chain32 != (int)value) { h->chain32 = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_BF16_STORE:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            if (h->bf16_store != (int)value) { h->bf16_store = (int)value; return drop_plans(h); }
            return FEAR_OK;
        case FEAR_OPT_SPLIT_STREAMS:
            if (value != 0 && value != 1) return FEAR_ERR_SHAPE;
            h->split_streams = (int)value;      // (same plans: only how a pass is issued changes)
            return FEAR_OK;
        default: return FEAR_ERR_SHAPE;
    }
}

int64_t fear_get_option(fear_handle* h, int option) {
    if (!h) return FEAR_ERR_NULL;
    switch (option) {
        case FEAR_OPT_MAX_BATCH: return h->max_batch;
        case FEAR_OPT_PROFILE: return h->profile;
        case FEAR_OPT_PROFILE_OP: return h->profile_op;
        case FEAR_OPT_FUSE: return h->fuse;
        case FEAR_OPT_MATH: return h->math;
        case FEAR_OPT_CHAIN: return h->chain;
        case FEAR_OPT_SMALL_PASS: return h->small_pass;
        case FEAR_OPT_PLAN_CROPS: return h->plan_crops;
        case FEAR_OPT_DUAL_HEAD: return h->dual_head;
        case FEAR_OPT_HEAD_STAGGER: return h->head_stagger_us;
        case FEAR_OPT_TILE_V4: return h->tile_v4;
        case FEAR_OPT_TINY_SEP: return h->tiny_sep;
        case FEAR_OPT_HEAD_CHAIN: return h->head_chain;
        case FEAR_OPT_E1_PAIR: return h->e1_pair;
        case FEAR_OPT_CHAIN32: return h->chain32;
        case FEAR_OPT_BF16_STORE: return h->bf16_store;
        case FEAR_OPT_SPLIT_STREAMS: return h->split_streams;
        default: return FEAR_ERR_SHAPE;
    }
}

// the plan a pass of nb crops runs on (build_plan's mode)
// ms per track call (profiles/r03_plan_sweep.txt), crops: tiny plan 1: 0.30  4: 0.34  8: 0.40  12: 0.50  16: 0.59  24: 0.78  32: 0.95
//                                                          small plan 1: 0.53  4: 0.54  8: 0.56  12: 0.59  16: 0.61  24: 0.68  32: 0.75
#ifndef FEAR_TINY_PASS
#define FEAR_TINY_PASS 16
#endif
constexpr int kTinyPass = FEAR_TINY_PASS;
static int small_pass(const fear_handle* h, int nb) { return h->fuse && nb <= h->small_pass ? (nb <= kTinyPass ? 2 : 1) : 0; }

// build_plan's mode for a pass of nb crops: small_pass's, or 3 for a MID-SIZE pass of the throughput plan — the chained 32 x 32 stage
// is one workgroup per crop for ~250 us whatever the crop count, while the four tile launches it replaces scale with it: up to 128
// crops the tiles win (1.507 vs 1.523 ms at 104 crops, 1.558 vs 1.572 at 128; 1.983 vs 1.940 at 143 / 144, 2.008 vs 1.964 at 160, 2.226 vs 2.167 at 256:
// profiles/r06_chain32_midsize.txt).  Only while the automatic plan selection is on (FEAR_OPT_SMALL_PASS > 0): with 0 the caller
// asks for THE throughput plan at every size (A/Bs, tests).
#ifndef FEAR_CHAIN32_MIN_CROPS
#define FEAR_CHAIN32_MIN_CROPS 129
#endif
static int plan_mode(const fear_handle* h, int nb) {
    const int sm = small_pass(h, nb);
    return sm == 0 && h->small_pass > 0 && h->chain32 && nb < FEAR_CHAIN32_MIN_CROPS ? 3 : sm;
}

// the plan fear_plan_* / fear_profile_read describe: that of a pass of FEAR_OPT_PLAN_CROPS crops (default: a full pass)
static int introspected_small(const fear_handle* h, bool with_head = true) {
    const int nb = h->plan_crops > 0 ? h->plan_crops : h->max_batch;
    return with_head ? plan_mode(h, nb) : small_pass(h, nb);
}

int fear_features(fear_handle* h, const float* img, int n, int hw, float* out, void* stream) {
    if (!h) return FEAR_ERR_NULL;
    if (n < 0) return FEAR_ERR_SHAPE;
    if (n == 0) return FEAR_OK;   // empty batch: nothing to read or write, null tensors are fine
    if (!img || !out) return FEAR_ERR_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    const int fhw = (hw / 16) * (hw / 16);
    for (int b0 = 0; b0 < n; b0 += h->max_batch) {
        const int nb = n - b0 < h->max_batch ? n - b0 : h->max_batch;
        Plan* p = nullptr;
        int st = build_plan(h, hw, false, small_pass(h, nb), &p);      // (the template branch has no mid-size variant)
        if (st != FEAR_OK) return st;
        st = ensure_workspace(h, *p);
        if (st != FEAR_OK) return st;
        Ext ext{};
        ext.img = img + (size_t)b0 * 3 * hw * hw;
        ext.feat_out = out + (size_t)b0 * h->feat_channels * fhw;
        st = run_plan(h, *p, nb, ext, static_cast<hipStream_t>(stream));
        if (st != FEAR_OK) return st;
    }
    return FEAR_OK;
}

static int track_impl(fear_handle* h, const float* search, const float* tmpl, const float* tmpl_cls, int n, float* bbox,
                      long bbox_stride, float* cls, long cls_stride, void* stream) {
    if (!h) return FEAR_ERR_NULL;
    if (n < 0) return FEAR_ERR_SHAPE;
    if (n == 0) return FEAR_OK;
    if (!search || !tmpl || !bbox || !cls) return FEAR_ERR_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    const int hw = 256;
    const size_t tz = (size_t)h->feat_channels * 64;
    for (int b0 = 0; b0 < n; b0 += h->max_batch) {
        const int nb = n - b0 < h->max_batch ? n - b0 : h->max_batch;
        Plan* p = nullptr;
        int st = build_plan(h, hw, true, plan_mode(h, nb), &p);
        if (st != FEAR_OK) return st;
        st = ensure_workspace(h, *p);
        if (st != FEAR_OK) return st;
        Ext ext{};
        ext.img = search + (size_t)b0 * 3 * hw * hw;
        ext.tmpl = tmpl + (size_t)b0 * tz;
        ext.tmpl_cls = tmpl_cls ? tmpl_cls + (size_t)b0 * tz : nullptr;
        ext.bbox_out = bbox + (size_t)b0 * bbox_stride;
        ext.cls_out = cls + (size_t)b0 * cls_stride;
        ext.bbox_stride = bbox_stride;
        ext.cls_stride = cls_stride;
        hipStream_t s_main = static_cast<hipStream_t>(stream);
        // FEAR_OPT_SPLIT_STREAMS: the pass as two half-batches, the second on the handle's own stream.  The crops of a batch are
        // independent and every kernel computes a crop the same way whatever its neighbours (tests: batch invariance, bit for bit),
        // so the maps are identical; what changes is that one half's launch boundaries, prologues and memory-side kernels fall into
        // the other half's ALU-bound stretches.  Only when both halves still run the throughput plan, and never while profiling
        // (per-op events assume one stream).
        const int nA = (nb + 1) / 2, nB = nb - nA;
        if (h->split_streams && !h->profile && small_pass(h, nb) == 0 && nB > 0 && small_pass(h, nB) == 0) {
            if (!h->split_stream) {
                HIP_TRY(h, hipStreamCreateWithFlags(&h->split_stream, hipStreamNonBlocking));
                HIP_TRY(h, hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
                HIP_TRY(h, hipEventCreateWithFlags(&h->split_join, hipEventDisableTiming));
            }
            // per-call bookkeeping of run_plan, done here around both halves: a previous call on another stream is waited for
            if (!h->stream_switch) HIP_TRY(h, hipEventCreateWithFlags(&h->stream_switch, hipEventDisableTiming));
            if (h->last_stream_valid && h->last_stream != s_main) HIP_TRY(h, hipStreamWaitEvent(s_main, h->stream_switch, 0));
            h->last_stream = s_main;
            h->last_stream_valid = true;
            HIP_TRY(h, hipEventRecord(h->split_fork, s_main));               // inputs ready, the workspace free
            HIP_TRY(h, hipStreamWaitEvent(h->split_stream, h->split_fork, 0));
            st = run_plan(h, *p, nA, ext, s_main, 0, true);
            int st_b = FEAR_OK;
            if (st == FEAR_OK) {
                Ext eb = ext;
                eb.img = ext.img + (size_t)nA * 3 * hw * hw;
                eb.tmpl = ext.tmpl + (size_t)nA * tz;
                eb.tmpl_cls = ext.tmpl_cls ? ext.tmpl_cls + (size_t)nA * tz : nullptr;
                eb.bbox_out = ext.bbox_out + (size_t)nA * bbox_stride;
                eb.cls_out = ext.cls_out + (size_t)nA * cls_stride;
                st_b = run_plan(h, *p, nB, eb, h->split_stream, (size_t)nA, true);
            }
            // join even when a half failed: whatever it did enqueue on the handle's stream still uses the shared workspace, and a
            // later call must not start before it has drained
            HIP_TRY(h, hipEventRecord(h->split_join, h->split_stream));      // the caller's stream continues once both halves are done
            HIP_TRY(h, hipStreamWaitEvent(s_main, h->split_join, 0));
            HIP_TRY(h, hipEventRecord(h->stream_switch, s_main));
            if (st == FEAR_OK) st = st_b;
        } else {
            st = run_plan(h, *p, nb, ext, s_main);
        }
        if (st != FEAR_OK) return st;
    }
    return FEAR_OK;
}

int fear_track(fear_handle* h, const float* search, const float* tmpl, const float* tmpl_cls, int n, float* bbox,
               float* cls, void* stream) {
    return track_impl(h, search, tmpl, tmpl_cls, n, bbox, 4 * 256, cls, 256, stream);
}

int fear_track_packed(fear_handle* h, const float* search, const float* tmpl, const float* tmpl_cls, int n, float* maps,
                      void* stream) {
    // bbox -> channels 0..3, cls -> channel 4 of one (n,5,16,16) tensor: the payload of the multi-GPU all-gather
    return track_impl(h, search, tmpl, tmpl_cls, n, maps, 5 * 256, maps ? maps + 4 * 256 : nullptr, 5 * 256, stream);
}

int fear_decode_smooth(fear_handle* h, const float* cls, const float* bbox, int n, int score_size, int total_stride,
                       int instance_size, const double* prev_size, const double* window, double penalty_k,
                       double window_influence, double lr, int32_t* rc, double* xywh, float* score, void* stream) {
    if (!h) return FEAR_ERR_NULL;
    if (n < 0 || score_size < 1 || score_size > 64) return FEAR_ERR_SHAPE;
    if (n == 0) return FEAR_OK;
    if (!cls || !bbox || !prev_size || !window || !rc || !xywh || !score) return FEAR_ERR_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    DecodeSmoothArgs a{cls, bbox, prev_size, window, rc, xywh, score, n, score_size, total_stride, instance_size,
                       penalty_k, window_influence, lr};
    hipLaunchKernelGGL(decode_smooth_kernel, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return FEAR_OK;
}

int fear_decode(fear_handle* h, const float* cls, const float* bbox, int n, int score_size, int total_stride,
                int instance_size, int32_t* rc, double* xywh, float* score, void* stream) {
    if (!h) return FEAR_ERR_NULL;
    if (n < 0 || score_size < 1 || score_size > 64) return FEAR_ERR_SHAPE;
    if (n == 0) return FEAR_OK;
    if (!cls || !bbox || !rc || !xywh || !score) return FEAR_ERR_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    DecodeArgs a{cls, bbox, rc, xywh, score, n, score_size, total_stride, instance_size};
    hipLaunchKernelGGL(decode_kernel, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return FEAR_OK;
}

int fear_normalize_u8(fear_handle* h, const uint8_t* u8, int n, int hw, float* out, void* stream) {
    if (!h) return FEAR_ERR_NULL;
    if (n < 0 || hw < 1) return FEAR_ERR_SHAPE;
    if (n == 0) return FEAR_OK;
    if (!u8 || !out) return FEAR_ERR_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    NormArgs a{};
    a.X = u8; a.Y = out; a.plane = hw * hw; a.pixels = (long)n * hw * hw;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c) {
        a.mean[c] = mean[c] * 255.0f;
        a.inv_std[c] = 1.0f / (stdv[c] * 255.0f);
    }
    hipLaunchKernelGGL(normalize_kernel, dim3((a.pixels + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return FEAR_OK;
}

int fear_crop_normalize(fear_handle* h, const uint8_t* frame_u8, int frame_h, int frame_w, const int32_t* ctx_xywh,
                        const uint8_t* pad_rgb, int n, int out_hw, float* out, void* stream) {
    if (!h) return FEAR_ERR_NULL;
    if (n < 0 || frame_h < 1 || frame_w < 1 || out_hw < 1 || out_hw > 4096) return FEAR_ERR_SHAPE;
    if (n == 0) return FEAR_OK;
    if (!frame_u8 || !ctx_xywh || !pad_rgb || !out) return FEAR_ERR_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    CropArgs a{};
    a.frame = frame_u8; a.ctx = ctx_xywh; a.pad = pad_rgb; a.out = out;
    a.H = frame_h; a.W = frame_w; a.S = out_hw; a.n = n;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c) {
        a.mean[c] = mean[c] * 255.0f;
        a.inv_std[c] = 1.0f / (stdv[c] * 255.0f);
    }
    const long total = (long)n * out_hw * out_hw;
    hipLaunchKernelGGL(crop_resize_normalize_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return FEAR_OK;
}

int fear_plan_size(fear_handle* h, int hw, int with_head) {
    if (!h) return FEAR_ERR_NULL;
    Plan* p = nullptr;
    int st = build_plan(h, hw, with_head != 0, introspected_small(h, with_head != 0), &p);
    if (st != FEAR_OK) return st;
    return (int)p->ops.size();
}

int fear_plan_op(fear_handle* h, int hw, int with_head, int i, char* name64, double* flops_per_crop,
                 double* bytes_per_crop) {
    if (!h) return FEAR_ERR_NULL;
    Plan* p = nullptr;
    int st = build_plan(h, hw, with_head != 0, introspected_small(h, with_head != 0), &p);
    if (st != FEAR_OK) return st;
    if (i < 0 || i >= (int)p->ops.size()) return FEAR_ERR_SHAPE;
    const Op& op = p->ops[i];
    if (name64) { strncpy(name64, op.name, 63); name64[63] = 0; }
    if (flops_per_crop) *flops_per_crop = op.flops;
    if (bytes_per_crop) *bytes_per_crop = op.bytes;
    return FEAR_OK;
}

int fear_profile_read(fear_handle* h, int hw, int with_head, int i, double* total_ms, int64_t* launches) {
    if (!h) return FEAR_ERR_NULL;
    Plan* p = nullptr;
    int st = build_plan(h, hw, with_head != 0, introspected_small(h, with_head != 0), &p);
    if (st != FEAR_OK) return st;
    if (i < 0 || i >= (int)p->ops.size()) return FEAR_ERR_SHAPE;
    st = drain_events(h);
    if (st != FEAR_OK) return st;
    if (total_ms) *total_ms = p->ops[i].prof_ms;
    if (launches) *launches = p->ops[i].prof_n;
    return FEAR_OK;
}

int fear_profile_reset(fear_handle* h) {
    if (!h) return FEAR_ERR_NULL;
    int st = drain_events(h);
    if (st != FEAR_OK) return st;
    for (auto& kv : h->plans)
        for (Op& op : kv.second->ops) { op.prof_ms = 0; op.prof_n = 0; }
    return FEAR_OK;
}

size_t fear_workspace_bytes(fear_handle* h) { return h ? h->workspace_floats * sizeof(float) : 0; }

int fear_last_hip_error(fear_handle* h) { return h ? h->last_hip_error : 0; }

}  // extern "C"

// the head training-step operators (include/fear_train.h) share this translation unit
#include "fear_train.hip"
