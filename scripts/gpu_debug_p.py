import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_binding as ob, synth
import thz_image_explorer_amd as pkg
from test_gpu_parity import gpu_fft_stage, phase_ok
eng = pkg.Engine(0)
for nt in (1000, 1001):
    nx, ny = 3, 7
    time, cube = synth.make_cube(nx, ny, nt)
    w = pkg.host_fft_window(time, 0, 1.0, 7.0)
    ref = ob.fft_stage(cube, time, 0, 1.0, 7.0)
    for fam in (0, 2):
        eng.set_kernel_family(fam); eng.set_time_axis(time)
        got = gpu_fft_stage(eng, cube, w)
        d = got["phases"].astype(np.float64) - ref["phases"]
        j = np.round(d / (2 * np.pi))
        res = np.abs(d - 2 * np.pi * j)
        print(nt, fam, eng.kernel_variant(), "ok" if phase_ok(got["phases"], ref["phases"], ref["amplitudes"]) else "FAIL",
              "max residual", res.max(), "at", np.unravel_index(res.argmax(), res.shape), "flips", np.argwhere(j != 0)[:6].tolist())
        k = np.unravel_index(res.argmax(), res.shape)
        print("   got", got["phases"][k], "ref", ref["phases"][k], "amp", ref["amplitudes"][k], "ampmax", ref["amplitudes"][k[0], k[1]].max(),
              "fft got", got["fft"][k], "ref", ref["fft"][k])
        if nt == 1000:
            print("   trace(0,1) bin0: got phase", got["phases"][0,1,0], "ref", ref["phases"][0,1,0], "got fft", got["fft"][0,1,0], np.signbit(got["fft"][0,1,0]),
                  "ref fft", ref["fft"][0,1,0], np.signbit(ref["fft"][0,1,0]), "bins1-3 got", got["phases"][0,1,1:4], "ref", ref["phases"][0,1,1:4])
