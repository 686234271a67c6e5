import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'smpl_skin' in n:
        g=int(r['Grid_Size_X'])//256
        nm=n[n.index('smpl_skin'):].split('(')[0]
        d[(nm, g)].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    v=sorted(v); print(k, len(v), 'median us', v[len(v)//2]/1e3, 'min', v[0]/1e3)
