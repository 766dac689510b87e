"""Per-phase s_memtime stamps of one workgroup of a tile-loop kernel (how the DESIGN.md 4 phase tables were made).

The kernels in the tree carry no instrumentation.  To time the phases of a kernel, temporarily
  * add `long long* dbg` to its argument struct, set from the environment variable GNET_DBG_PTR at the launch site
    (`strtoull(getenv("GNET_DBG_PTR"), 0, 10)`),
  * define `#define STAMP(k) do { if (a.dbg && blockIdx.x == 8 && tid == 0 && it < 40) a.dbg[it * 16 + (k)] = clock64(); } while (0)`
    and put STAMP(0..K-1) at the phase boundaries of the tile loop (`it` = tile counter of the workgroup),
  * rebuild and run  NSTAMP=K python tools/phase_stamps.py  on the GPU.
clock64() is s_memtime = core clock cycles (calibrated against a pure-MFMA loop in tools/mfma_valu_overlap.hip).
"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(40 * 16, dtype=torch.int64, device="cuda")
os.environ["GNET_DBG_PTR"] = str(dbg.data_ptr())
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
experiment_cfg()
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
