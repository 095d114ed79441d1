import numpy as np, torch, sys
sys.path.insert(0, ".")
from gaussianprediction_amd.weights_ops import knn_keypoints
from oracle import weights_oracle as wo
rng = np.random.default_rng(7)
N, K, nn = 40000, 250, 6
xyz = torch.tensor(rng.uniform(-1.3, 1.3, size=(N, 3)).astype(np.float32)).cuda()
feat = torch.tensor((1e-3 * rng.uniform(-1, 1, size=(N, 32))).astype(np.float32)).cuda()
kp = xyz[torch.tensor(rng.choice(N, K, replace=False)).cuda()].clone()
kp[:10] = kp[10:20]
kpf = feat[:K].clone()
kpf[:10] = kpf[10:20]
print(torch.equal(kp[:10], kp[10:20]), torch.equal(kpf[:10], kpf[10:20]))
bi, bd = knn_keypoints(xyz, kp, nn, feat, kpf, 5.0, "hybird", return_dist=True)
X = np.concatenate([xyz.cpu().numpy(), np.float32(5.0) * feat.cpu().numpy()], 1)[:2000]
Kp = np.concatenate([kp.cpu().numpy(), np.float32(5.0) * kpf.cpu().numpy()], 1)
ri, rd = wo.knn(X, Kp, nn)
bad = np.nonzero((bi[:2000].cpu().numpy() != ri).any(1))[0]
print(len(bad), bad[:5])
r = bad[0]
print(bi[r].tolist(), bd[r].tolist()); print(ri[r].tolist(), rd[r].tolist())
