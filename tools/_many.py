import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from visual_slam_amd import orbhip, synth
from oracle import orb_oracle as O
w, h, n = 96, 96, 70000
prm = (200, 1.2, 3, 20, 7)
dev = torch.device('cuda', 0)
base = np.stack([synth.frame('blocks', w, h, i) for i in range(64)])
frames = torch.from_numpy(base).to(dev).repeat((n + 63) // 64, 1, 1)[:n].contiguous()
g = orbhip.ORBextractor(*prm, max_batch=100000)
cap = g.max_keypoints(w, h)
kp = torch.empty((n, cap, 7), dtype=torch.float32, device=dev); de = torch.empty((n, cap, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
t0 = time.time()
g.extract_batch_device(frames.data_ptr(), n, w, h, w, w * h, kp.data_ptr(), de.data_ptr(), cnt.data_ptr(), cap)
g.synchronize(); torch.cuda.synchronize()
print('extracted', n, 'frames in %.3f s' % (time.time() - t0))
c = cnt.cpu().numpy()
ok = True
for i in (0, 1, 63, 64, 65534, 65535, 65536, 69999):
    o = O.OracleExtractor(*prm); okp, od = o.extract(base[i % 64])
    k = kp[i, :c[i]].cpu().numpy()
    good = c[i] == len(okp) and np.array_equal(k[:, 0], okp['x']) and np.array_equal(k[:, 1], okp['y']) and np.array_equal(de[i, :c[i]].cpu().numpy(), od)
    ok &= bool(good); print(i, c[i], len(okp), good)
print('OK' if ok else 'MISMATCH')
