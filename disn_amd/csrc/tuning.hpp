// Tuning / ablation knobs.  The product library (disn_amd/csrc/build.py, default) has NONE: every knob
// is a compile-time constant here and no environment variable is read anywhere.  `build.py --tuning`
// compiles the same sources with -DDISN_TUNING into libdisn_amd_tuning.so, where the knobs are
// run-time integers behind the extra export disn_tuning_set(); only tools/ load that library.
#pragma once

namespace disn {
namespace tune {
#ifdef DISN_TUNING
extern int x3;          // 0: f32-input MFMA everywhere (no three-term bf16 kernels)
extern int overlap;     // 0: disn_encode_query on the caller's stream only
extern int bf_splits;   // > 0: force this split-K factor in the three-term kernels
extern int skip_pack;   // 1: disn_conv3x3_bf16 reuses the packed image of the previous call (timing only)
extern int fused_safe;  // 1: fused point MLP waits for ALL LDS-DMA at every sync (debugging)
extern int gemm_force[3];  // {BM, BN, workgroups}: plan of the f32-input GEMM when BM != 0 (tools/sweep_gemm.py)
extern int gemv_wgs;     // > 0: workgroups of the split-K GEMV (default 2048 = every wave slot of the chip)
extern int dense_mb;     // > 0: row blocks (of 32) per dense_h2 tile: 1, 2 or 4 (tools/dense_h2_time.py)
extern int dense_nw;     // 4: 128-column dense_h2 tiles (16 waves) with dense_mb = 2
extern int dense_kpw;    // 4: 256-column chunks with dense_nw = 4
extern int conv_occ;     // 1: never the two-workgroups-per-CU conv_h2 variants (tools/conv_stack_time.py)
extern int conv_occ_mask;  // which tilings get them (bit 0: <1,1,16,14>, bit 1: <2,2,32,28>, bit 2: <4,2,16,16> -> <2,2,16,16>)
extern int conv_occ_min; // from this many workgroups per launch on
extern int aux_cu_mode;  // > 0: disn_ctx_create puts the auxiliary stream on a CU subset (hipExtStreamCreateWithCUMask)
extern int conv_img_major;  // 0 / 1: force conv_h2's tile order (n-tile-major / image-major); -1: by shape
extern int conv_wide_min;  // images per call from which conv_h2_launch takes conv_h2w.hip (1 << 30: never)
extern int l4_ranges;     // retired (round 4): the K ranges of the local fold2/conv1 of a batched call went with the fused small-set kernels
extern int gather_l16;    // 1: project_gather_taps_kernel issues its 16 tap loads before using any (0, default: tap_pixel by tap_pixel --
                          // measured FASTER: 84 vs 105 us for 8 x 2048 points, profiles/r03h_gather_time.txt)
extern int tn_interleave;  // -1: by form and tile count (default); 0 / 1: never / always interleaved row steps in the weight-gradient GEMM
                           // (gemm_tn_mfma.hip), > 1: that many workgroups per tile
extern int conv5_whole;    // 0: never the whole-image tiling of 14 x 14 layers (conv_h2_launch)
extern int fused_small;    // 0: batched calls keep round 3's layer-by-layer point MLP (dense_h2w.hip) instead of the fused small-set kernels
extern int conv11_wgs;     // > 0: workgroups of conv1_1_direct_kernel's persistent grid (default 512)
extern int conv11_rt;      // > 0: row pairs per tile of conv1_1_direct_kernel in a batched call (1, 2, 4; default 4)
extern int gemv_rows_cfg;   // one-row gemv_rows_kernel<1, R, U>: 0 by N (default), 1 <1,8>, 2 <2,8>, 3 <1,16>, 4 <4,4>, 5 <2,4>, 6 <1,4> (same bits: the k order of a lane is U-independent)
extern int densew_m64, densew_c128;   // >= 0: force dense_h2w's 64-row tiles / forbid its 128-column chunks
extern long long* ch2_stamps;  // conv_h2 kernels write 16 clock stamps per workgroup here (tools/conv_h2_stamps.py)
#else
constexpr int x3 = 1, overlap = 1, bf_splits = 0, skip_pack = 0, fused_safe = 0, gemv_wgs = 0, dense_mb = 0, dense_nw = 0, dense_kpw = 0, conv_occ = 0, conv_occ_mask = 7 /* 15 with the conv4 variant */, conv_occ_min = 384, conv_img_major = -1, conv_wide_min = 4, l4_ranges = 2, gather_l16 = 0, densew_m64 = -1, densew_c128 = -1, conv11_wgs = 0, conv11_rt = 0, conv5_whole = 1, tn_interleave = -1, fused_small = 1, gemv_rows_cfg = 0;
constexpr int gemm_force[3] = {0, 0, 0};
#endif
}  // namespace tune
}  // namespace disn
