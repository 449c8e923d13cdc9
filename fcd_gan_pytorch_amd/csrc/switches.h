// The ONE table of run-time switches of libfcdgan_hip.so and of the Python host code above it.
//
// Every switch is an integer.  The table is filled ONCE, when the library is loaded, from the environment variables
// FCD_<NAME>; nothing reads the environment after that (no getenv on any call path).  A/B runs and tests change a switch at
// run time through the C ABI -- fcd_switch_set("WGRAD_SPLIT", 0) -- and README.md's list of switches is printed from this
// table (tools/list_switches.py -> fcd_switch_count / fcd_switch_name / fcd_switch_default / fcd_switch_help).
// Defaults are what every reported number uses; a switch changes HOW a result is computed (which kernel, which launch
// geometry), never WHAT is computed, unless its help text says "experimental".
#pragma once

// X(NAME, default, help)
#define FCD_SWITCH_TABLE(X)                                                                                                        \
  /* ---- layer plans */                                                                                                           \
  X(WINO, 4, "Winograd tile size of the planned 3x3 layers: 4 = F(4x4,3x3), 2 = F(2x2,3x3), 0 = direct kernels only (also fcd_conv_wino_set)") \
  X(WINO_MINC, 64, "F(4x4) plan: least reduction channels")                                                                         \
  X(WINO_MINROWS, 128, "F(4x4) plan: least GEMM rows")                                                                              \
  X(WINO_WG_MINK, 128, "F(4x4) weight-gradient plan: least filters")                                                                \
  X(WINO_WG_MINC, 128, "F(4x4) weight-gradient plan: least channels")                                                               \
  X(WINO_WG_WGS, 0, "F(4x4) weight gradient: reduction splits = ceil(n / blocks) (round-4 rule, n = 1024); 0 = the round model")    \
  X(WINO2, 1, "fused F(2x2,3x3) kernel for the 64-row 3x3 layers (0: those layers on the direct kernels)")                          \
  X(WINO2_MINC, 32, "fused F(2x2): least reduction channels")                                                                       \
  X(WINO2_WAVES, 8, "fused F(2x2): 8 = one 8x32 workgroup per CU, 4 = two 4x32 workgroups, 1 = one wave per SIMD")                   \
  X(WINO2_KS, 2, "fused F(2x2): channels per LDS stage / 4 (1, 2 or 3)")                                                            \
  /* ---- F(4x4) GEMM */                                                                                                            \
  X(WINO_SPLIT, 1, "F(4x4) GEMMs on the bf16 matrix pipe with exact three-way operand splitting (0: v_mfma_f32_32x32x2_f32; also fcd_conv_wino_split_set)") \
  X(WINO_SPLIT_BIG, 1, "split GEMM tile policy: 0 = 128x128 only, 1 = 256x256 where it fills the chip, 2 = 256x256 for every GEMM with >= 256 rows") \
  X(WINO_TILE, 0, "fp32-pipe GEMM tile: 0 = 128x128, 1 = 256x128, 2 = 256x256")                                                    \
  X(WINO_XB, 0, "transform positions chained per GEMM workgroup (0: chosen per launch)")                                            \
  X(WINO_XCD2, 128, "largest tile count per transform-position group whose tiles all go to one XCD (0: never)")                     \
  X(WINO_RES, 1, "filter-resident split GEMM for one-row-tile layers with a short reduction")                                       \
  X(WINO_RES_WGS, 256, "workgroups in flight of the filter-resident GEMM")                                                          \
  X(WINO_CBLK, 1, "GEMM result in 32x32 MFMA-native blocks (0: row-major M)")                                                       \
  /* ---- F(4x4) transforms */                                                                                                      \
  X(WINO_IN_ROLL, 4, "strips per block of the rolling input transform (1: one-strip kernel)")                                       \
  X(WINO_IN_EXP, 0, "input-transform launch experiments (bit 3: no XCD-aware order)")                                               \
  X(WINO_BNSTATS, 1, "BatchNorm statistics out of the F(4x4) output transform")                                                     \
  X(WINO_KEEPV, 1, "weight gradient reuses the forward pass's transformed input (0: re-transforms x)")                              \
  X(WINO_CHAIN, 1, "frozen VGG runs through the fused output->input transform (0: layer by layer)")                                 \
  X(WINO_RELU_BITS, 1, "frozen F(4x4) layers keep 16 sign bits per tile as the ReLU mask (0: fp32 y on the tape)")                  \
  X(BN_FUSE, 1, "train-mode BatchNorm + ReLU inside the next convolution's loader / the density head (0: kernels of their own)")    \
  X(PAIR_CAT, 1, "first decoder convolution reads skip pair + upsampled tensor in place (0: materialise the concatenation)")        \
  /* ---- direct convolution */                                                                                                     \
  X(CONV_XCD, 1, "XCD-aware workgroup -> tile order")                                                                               \
  X(CONV_BIG, 1, "128x256 direct tiles")                                                                                            \
  X(CONV_THINFWD, 1, "thin-channel forward kernel for <= 4 input channels")                                                         \
  X(CONV_ROWS16, 1, "16-row MFMA tiles for <= 16-filter 9x9 layers")                                                                \
  X(CONV_HEAD, 1, "one-filter 1x1 head kernels (0: generic 1x1 path + ATen sigmoid)")                                               \
  X(S2_SUBPIXEL, 1, "stride-2 data gradient as four sub-pixel convolutions (0: zero-dilated)")                                      \
  X(S2_GLDS, 1, "its LDS-DMA kernel (0: register-staged kernel for every layer)")                                                   \
  X(THIN_MFMA, 32, "rows per workgroup of the matrix-core data gradient of the one-band first VGG layer (0: VALU kernel)")          \
  /* ---- direct weight gradient */                                                                                                 \
  X(WGRAD_WGS, 512, "workgroups the reduction split of the direct weight gradient aims at")                                         \
  X(WGRAD_TKC, 1, "split partials in (tap, filter, channel) order (0: dW's own layout)")                                            \
  X(WGRAD_ROLL, 1, "rolling 4-row ring kernel for 3x3 stride-1 layers")                                                             \
  X(WGRAD_NCHW, 1, "3x3 weight gradient reads x and dY as they lie (0: channel-minor copies first)")                                \
  X(WGRAD_SPLIT, 1, "that kernel on the bf16 matrix pipe, operands split exactly in three (0: fp32 pipe, stride 1 only; also fcd_conv_wgrad_split_set)") \
  X(WGRAD_THIN, 1, "first-layer weight-gradient kernel (<= 4 channels)")                                                            \
  X(WGRAD_THIN9, 1, "9x9 weight-gradient kernel with <= 4 channels on one side")                                                    \
  /* ---- BatchNorm / pooling / resize */                                                                                           \
  X(BN_POOL, 1, "encoder tails: BatchNorm + ReLU + max-pool + skip-gradient sum as one node (0: separate kernels)")                 \
  X(UPSAMPLE_ROWS, 1, "row-walking kernel for the adjoint of the x2 upsampling")                                                    \
  X(NO_POOLFUSE, 0, "1: ReLU + max-pool behind a frozen convolution as kernels of their own")                                       \
  /* ---- host code (the Python modules read these through fcd_switch_get) */                                                \
  X(PACK_MULTI, 1, "all F(4x4) filter packs of an optimizer re-packed by one launch after its step")                                \
  X(FUSED_GLUE, 1, "masked stacks / perception-tap MSE as fused kernels (0: ATen compositions)")                                    \
  X(D_SHARE, 1, "Demo_RSSS Discriminator step: the shared masked x through D's net once (0: twice, as the reference)")              \
  X(D_POOL, 0, "Discriminator pooled pair difference: 0 = fused kernel, 1 = ATen (f_x - f_y).mean(), 2 = mean first (round 3)")     \
  X(LAUNCH_WINDOW, 384, "host: launches the issuing thread may be ahead of the device before it sleeps (0: run ahead until the hardware queue is full and spin there)") \
  X(STEP_OVERLAP, 0, "experimental: the Discriminator step of the adversarial loops on a second HIP stream")                        \
  X(DP_FORCE_EXCHANGE, 0, "one-rank process groups run every data-parallel collective")

enum {
#define FCD_SW_ENUM(NAME, DEF, HELP) FCD_SW_##NAME,
  FCD_SWITCH_TABLE(FCD_SW_ENUM)
#undef FCD_SW_ENUM
      FCD_SW_COUNT
};

extern int g_fcd_switch[FCD_SW_COUNT];      // common.hip; filled by a load-time constructor
static inline int fcd_sw(int id) { return g_fcd_switch[id]; }
