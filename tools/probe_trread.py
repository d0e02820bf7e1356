"""Print the lane/element mapping of ds_read_b64_tr_b16 on this GPU (used to design the wgrad LDS layout)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
lane = np.arange(64)
addr = ((lane >> 4) * 2048 + (lane & 15) * 64).astype(np.int32)     # lane i of group g -> chunk at g*2048 + i*64
a = torch.from_numpy(addr).cuda(); out = torch.zeros(256, dtype=torch.int16, device="cuda")
L.check(lib.hulc_k_trread_probe(a.data_ptr(), out.data_ptr(), None)); torch.cuda.synchronize()
o = out.cpu().numpy().astype(np.int64).reshape(64, 4) & 0xFFFF
for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 33, 63):
    desc = []
    for j in range(4):
        v = int(o[l, j]); g = v // 2048; i = (v % 2048) // 64; e = v % 64
        desc.append(f"(grp{g} lane{i} elem{e})")
    print(f"lane {l:2d}: " + " ".join(desc))
# check hypothesis: result[i][j] = chunk_of_lane[j*4 + i//4][i%4] within the group
ok = all(int(o[l, j]) == (l >> 4) * 2048 + (j * 4 + (l & 15) // 4) * 64 + (l & 15) % 4 for l in range(64) for j in range(4))
print("hypothesis result[i][j] = lane(j*4+i/4).elem(i%4):", ok)
