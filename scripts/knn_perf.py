"""Times the distance kernel alone on device-resident images (HIP events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sfm_amd import matcher, synth

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 32768]
    ctx = matcher.Context(0)
    ctx.profile_enable(True)
    for n in sizes:
        Q, T, _ = synth.planted_pair(n, n, 1001, 1002, 1003)
        iq = matcher.Image(ctx, Q); it = matcher.Image(ctx, T)
        out = torch.empty(n * 2 * 16, dtype=torch.uint8, device="cuda")
        cnt = torch.empty(n, dtype=torch.int32, device="cuda")
        for variant in [int(v) for v in os.environ.get('VARIANTS','0,1,2,3').split(',')]:
            ctx._L.sfm_gpu_debug_set_knn_variant(ctx.handle, variant)
            for _ in range(3):
                matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))
            ctx.synchronize(); ctx.profile_reset()
            reps = 20
            for _ in range(reps):
                matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))
            ms, k = ctx.profile_query("knn2_mfma")
            msf, _ = ctx.profile_query("knn_finalize")
            ops = 2.0 * n * n * 128
            t = ms / k * 1e-3
            rc = np.zeros(1, np.int32)
            ctx._L.sfm_gpu_debug_redo_count(ctx.handle, rc.ctypes.data)
            msr, kr = ctx.profile_query("knn_redo")
            print("   redo queries: %d  redo kernel %.1f us" % (rc[0], msr / max(kr, 1) * 1e3))
            print("n=%d variant=%d knn2_mfma %.1f us  %.1f TOP/s (%.1f%% of 5000)  finalize %.1f us" %
                  (n, variant, t * 1e6, ops / t / 1e12, ops / t / 5e15 * 100, msf / k * 1e3), flush=True)
        iq.close(); it.close()
main()
