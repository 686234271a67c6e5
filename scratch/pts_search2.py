import sys; sys.path.insert(0, "scripts")
from wino_points import *
torch.set_num_threads(8)
cands = {}
for a in [Fr(5,8), Fr(11,16), Fr(3,4), Fr(21,32), Fr(23,32)]:
    for b in [Fr(3,2), Fr(11,8), Fr(23,16), Fr(25,16), Fr(13,8)]:
        cands[f"0,+-{a},+-{b}"] = [0, a, -a, b, -b]
cands["lavin"] = [0,1,-1,2,-2]
data = [test_data(s, 256, 256, 14) for s in range(3)] + [test_data(7+s, 128, 128, 28) for s in range(2)] + [test_data(11+s, 512, 512, 7) for s in range(2)]
res = []
for name, pts in cands.items():
    AT, G, BT = cook_toom(pts); check_exact(AT, G, BT)
    acc = np.zeros(4)
    for x, w in data: acc += np.array(layer_error(AT, G, BT, x, w))
    res.append((acc[0]/acc[2], acc[1]/acc[3], name))
for r in sorted(res): print("%.2f %.2f %s" % r)
