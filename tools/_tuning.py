"""Knobs of the tuning build (disn_amd/csrc/tuning.hpp).  Tools only: build it with
`python -m disn_amd.csrc.build --tuning` and run the tool with
DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so (the product library has no knobs)."""
import ctypes as C

KEYS = {"x3": 0, "overlap": 1, "bf_splits": 2, "skip_pack": 3, "fused_safe": 4,
        "gemm_bm": 5, "gemm_bn": 6, "gemm_wgs": 7, "gemv_wgs": 8, "dense_mb": 9, "dense_nw": 10, "dense_kpw": 11, "conv_occ": 12, "conv_occ_mask": 13, "conv_occ_min": 14, "aux_cu_mode": 15, "conv_img_major": 16, "conv_wide_min": 17, "l4_ranges": 18, "gather_l16": 19, "densew_m64": 20, "densew_c128": 21, "conv11_wgs": 22, "tn_interleave": 23, "conv5_whole": 24, "fused_small": 25, "conv11_rt": 26, "gemv_rows_cfg": 27}


def set_knob(name: str, value: int) -> None:
    from disn_amd import _lib
    h = _lib.lib()
    try:
        fn = h.disn_tuning_set
    except AttributeError:
        raise SystemExit("this tool needs the tuning build: python -m disn_amd.csrc.build --tuning and "
                         "DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so")
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_int]
    if fn(KEYS[name], int(value)) != 0:
        raise ValueError(name)


def gemm_force(bm: int = 0, bn: int = 0, wgs: int = 0) -> None:
    set_knob("gemm_bn", bn)
    set_knob("gemm_wgs", wgs)
    set_knob("gemm_bm", bm)
