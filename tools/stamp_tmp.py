import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(40 * 16, dtype=torch.int64, device="cuda")
os.environ["GNET_DBG_PTR"] = str(dbg.data_ptr())
from gossipnet_amd.config import cfg, reset_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
reset_cfg()
net = Gnet(80, device=torch.device("cuda"))
imgs = [make_image(2000, 80, seed=i, preset="dense") for i in range(8)]
b = DeviceBatch(imgs, torch.device("cuda"))
for _ in range(3): net.run(b)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(40, 16)
names = sys.argv[1:]
K = int(os.environ.get("NSTAMP", "13"))
rows = [r for r in d if r[0] > 0 and r[K-1] > 0]
diffs = np.array([[r[k+1] - r[k] for k in range(K-1)] for r in rows[2:]], dtype=np.float64)
print("tiles", len(rows), "mean cycles per phase:", np.round(diffs.mean(0)).astype(int).tolist(), "total", int(diffs.sum(1).mean()))
gaps = [rows[i+1][0] - rows[i][K-1] for i in range(2, len(rows)-1)]
print("gap between tiles", np.mean(gaps))
