/*
 * fearw_format.h — on-disk layout of a `.fearw` FEAR model file (little endian).
 *
 * Written by tools/mlmodel_to_fearw.py from the BN-folded CoreML graphs the reference ships
 * (evaluate/coreml_convert.py:60-70 is what produced them); read by the HIP engine
 * (feartracker_amd/csrc/fear_model.cpp) and, independently, by the CPU oracle
 * (oracle/fear_oracle.py).
 *
 *   [FearwHeader 64 B][FearwConv x n_convs][FearwBlock x n_blocks][payload]
 *
 * Payload holds IEEE fp16 values exactly as stored in the .mlmodel (payload_dtype 0) — or fp32 values
 * (payload_dtype 1: weights exported from a training state, feartracker_amd/export.py) —: conv weights in
 * [Cout][Cin/groups][kH][kW] order followed by Cout biases when has_bias is set.
 * Offsets are bytes from the start of the payload, 16-byte aligned per conv.
 */
#ifndef FEARW_FORMAT_H
#define FEARW_FORMAT_H

#include <stdint.h>

#define FEARW_MAGIC "FEARW1\0\0"
#define FEARW_VERSION 1u

enum FearwBlockKind {
    FEARW_STEM = 0, /* conv[0] = 3x3 s2 conv + ReLU (fbnet_c stages[0])                        */
    FEARW_IR   = 1, /* conv[0] = 1x1 expand(+ReLU) or -1, conv[1] = depthwise(+ReLU),          */
                    /* conv[2] = 1x1 project (linear); residual: out += block input            */
    FEARW_NECK = 2, /* conv[0] = 1x1, linear (AdjustLayer, model/blocks.py:75-88)              */
    FEARW_SEP  = 3  /* conv[0] = depthwise 3x3 (linear), conv[1] = 1x1 followed by `act`       */
                    /* (SepConv, model/blocks.py:45-72, + folded BN)                           */
};

/* role of a FEARW_SEP block inside BoxTower (model/blocks.py:143-168) */
enum FearwRole {
    FEARW_ROLE_NONE = 0,
    FEARW_CLS_ENCODE = 1, FEARW_REG_ENCODE = 2, /* MatrixMobile.matrix11_s                      */
    FEARW_CLS_CORR = 3,   FEARW_REG_CORR = 4,   /* MobileCorrelation.enc on cat[x, z^T x]        */
    FEARW_BBOX_TOWER = 5, FEARW_CLS_TOWER = 6,
    FEARW_BBOX_PRED = 7,  FEARW_CLS_PRED = 8
};

enum FearwPayload { FEARW_PAYLOAD_F16 = 0, FEARW_PAYLOAD_F32 = 1 };

enum FearwAct { FEARW_ACT_NONE = 0, FEARW_ACT_RELU = 1, FEARW_ACT_EXP = 2 };

typedef struct FearwHeader {
    char     magic[8];
    uint32_t version;
    uint32_t n_convs;
    uint32_t n_blocks;
    uint32_t payload_dtype; /* 0 = fp16, 1 = fp32 (FearwPayload) */
    uint64_t payload_bytes;
    uint8_t  reserved[32];
} FearwHeader; /* 64 bytes */

typedef struct FearwConv {
    uint32_t cout, cin_per_group, groups, k, stride, pad;
    uint32_t relu;     /* a ReLU directly follows this conv */
    uint32_t has_bias;
    uint64_t w_off, b_off;
    char     name[24]; /* CoreML output tensor name, for diagnostics */
} FearwConv; /* 72 bytes */

typedef struct FearwBlock {
    uint32_t kind, role;
    int32_t  conv[3];
    uint32_t residual;
    uint32_t act;      /* activation after the block's last conv (FearwAct) */
    uint32_t reserved;
} FearwBlock; /* 32 bytes */

#endif /* FEARW_FORMAT_H */
