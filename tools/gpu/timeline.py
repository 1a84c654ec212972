import csv, sys, glob
d=sys.argv[1]
rows=[]
for f in glob.glob(d+'/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:40]))
for f in glob.glob(d+'/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')+' '+r.get('Bytes','')))
rows.sort()
# find the last-but-3 k_select and print until the next
idx=[i for i,r in enumerate(rows) if 'k_select' in r[2]]
a,b=idx[-4],idx[-3]
t0=rows[a][1]
prev=t0
for s,e,n in rows[a:b+1]:
    print('%9.1f us  +gap %6.1f  dur %7.1f  %s'%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,n)); prev=e
