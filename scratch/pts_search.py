import sys; sys.path.insert(0, "scripts")
from wino_points import *
import numpy as np
torch.set_num_threads(8)
from fractions import Fraction as Fr
cands = {}
for a in [Fr(1,2), Fr(4,7), Fr(3,5), Fr(5,8), Fr(2,3), Fr(7,10), Fr(5,7), Fr(3,4), Fr(4,5), Fr(5,6), Fr(7,8)]:
    for b in [1/a, Fr(1), Fr(5,4), Fr(4,3), Fr(3,2), Fr(7,4), Fr(2)]:
        if b == a: continue
        cands[f"0,+-{a},+-{b}"] = [0, a, -a, b, -b]
data = [test_data(s, 256, 256, 14) for s in range(2)] + [test_data(7, 128, 128, 28)]
res = []
for name, pts in cands.items():
    AT, G, BT = cook_toom(pts); check_exact(AT, G, BT)
    acc = np.zeros(4)
    for x, w in data: acc += np.array(layer_error(AT, G, BT, x, w))
    res.append((acc[0]/acc[2], acc[1]/acc[3], name))
for r in sorted(res)[:15]: print("%.2f %.2f %s" % r)
print("lavin", [r for r in res if r[2] == "0,+-1,+-2"])
