"""ctypes binding of tools/probe/libgpvprobe.so (gpv_probe.h): instruction-rate microbenchmarks, the shader-clock sampler and the
MFMA feasibility probe. Measurement only -- not part of the product package; bench.py and tools/mfma_probe.py import it by path."""
import ctypes
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libgpvprobe.so"
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError("%s not built -- make -C tools/probe" % LIB_PATH)
        try:
            import torch  # noqa: F401  (one HIP runtime per process, see gnark-plonky2-verifier_amd/_lib.py)
        except Exception:
            pass
        L = ctypes.CDLL(str(LIB_PATH))
        vp, sz, i32, dp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
        L.gpvp_microbench.argtypes = [i32, i32, dp]
        L.gpvp_microbench_clocked.argtypes = [i32, i32, dp, dp]
        L.gpvp_row_mix_rate.argtypes = [i32, i32, i32, dp]
        L.gpvp_clock_sample_begin.argtypes = [i32, ctypes.c_uint]
        L.gpvp_clock_sample_end.argtypes = [dp]
        L.gpvp_mfma_probe.argtypes = [i32, i32, vp, vp, vp, vp, sz, i32, dp]
        L.gpvp_mfma_probe_permute.argtypes = [i32, i32, vp, vp, sz, vp, sz, i32, dp]
        L.gpvp_mfma_probe_overlap.argtypes = [i32, i32, dp, vp, sz]
        L.gpvp_stream_write.argtypes = [i32, i32, sz, sz, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, dp]
        L.gpvp_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libgpvprobe error %d: %s" % (rc, lib().gpvp_last_error().decode("utf-8", "replace")))


MICROBENCH_NAMES = ["v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_add_co_u32", "v_mad_u32_u24", "v_add_u32",
                    "v_lshl_add_u64", "fr_row_mix(v_mad_u64_u32)"]


def microbench(which, device=0):
    v = ctypes.c_double()
    check(lib().gpvp_microbench(device, which, ctypes.byref(v)))
    return v.value


def microbench_clocked(which, device=0):
    """(lane-ops/s, shader clock in GHz during the timed launch)"""
    v, g = ctypes.c_double(), ctypes.c_double()
    check(lib().gpvp_microbench_clocked(device, which, ctypes.byref(v), ctypes.byref(g)))
    return v.value, g.value


def row_mix_rate(chains, waves, device=0):
    v = ctypes.c_double()
    check(lib().gpvp_row_mix_rate(device, chains, waves, ctypes.byref(v)))
    return v.value


def stream_write(store, n_streams, stride_words, steps, lanes_per_stream, chunk_words, spin, device=0):
    """ms of the output-stream write pattern (gpv_probe.h gpvp_stream_write)"""
    v = ctypes.c_double()
    check(lib().gpvp_stream_write(device, int(store), n_streams, stride_words, steps, lanes_per_stream, chunk_words, spin, ctypes.byref(v)))
    return v.value


def clock_sample_begin(microseconds, device=0):
    check(lib().gpvp_clock_sample_begin(device, int(microseconds)))


def clock_sample_end():
    v = ctypes.c_double()
    check(lib().gpvp_clock_sample_end(ctypes.byref(v)))
    return v.value


if __name__ == "__main__":
    print("# tools/probe/gpv_probe.py: instruction-rate microbenchmarks (lane-ops/s, whole chip)")
    print("# instruction                lane-ops/s   clock GHz   SIMD cycles per wave64 instruction (1024 SIMDs)")
    for i, nm in enumerate(MICROBENCH_NAMES):
        r, g = microbench_clocked(i)
        print("%-28s %.3e   %.3f       %.2f%s" % (nm, r, g, 1024 * g * 1e9 * 64 / r, "  (per multiply-add; the mix issues 41 instructions per 32 multiply-adds)" if i == 8 else ""))
    print("# Fr-row instruction mix: chains per lane x resident waves per SIMD -> multiply-adds/s (whole chip)")
    for waves in (1, 2, 3, 4):
        for chains in (1, 2):
            print("chains %d  waves/SIMD %d   %.3e" % (chains, waves, row_mix_rate(chains, waves)))
