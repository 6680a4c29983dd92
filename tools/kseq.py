import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[(r['Kernel_Name'][:34], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
idx=[i for i,n in enumerate(names) if 'k_snet4' in n[0]]
import collections
acc=collections.defaultdict(list)
for i0 in idx[4:12]:
    for j in range(i0, i0+9):
        acc[j-i0].append(names[j])
for k in sorted(acc):
    print(k, acc[k][0][0], round(sum(d for _,d in acc[k])/len(acc[k]),1))
