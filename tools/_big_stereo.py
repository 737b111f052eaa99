import sys, time, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_stereo import run_pair
cases = [('blocks', 3840, 2160, (8000, 1.2, 8, 20, 7)), ('blocks', 8192, 2048, (12000, 1.2, 8, 20, 7)), ('blocks', 16384, 512, (6000, 1.2, 8, 20, 7)),
         ('lowc', 16000, 300, (3000, 1.1, 12, 20, 7)), ('noise', 4096, 3072, (20000, 1.2, 6, 20, 7))]
for i in [int(a) for a in sys.argv[1:]]:
    kind, w, h, prm = cases[i]
    t0 = time.time()
    try:
        (okl, ou, od, on), (gkl, gu, gd, gn) = run_pair(kind, w, h, 0, prm)
    except Exception as ex:
        print(cases[i], 'FAILED:', type(ex).__name__, ex, flush=True); continue
    ok = len(okl) == len(gkl) and gn == on and np.array_equal(ou.view(np.uint32), gu.view(np.uint32)) and np.array_equal(od.view(np.uint32), gd.view(np.uint32))
    print(cases[i], 'keypoints', len(gkl), 'matches', gn, on, 'OK' if ok else 'MISMATCH', '%.1fs' % (time.time() - t0), flush=True)
