import csv,glob,collections,sys
f=glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace_stream") + "/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
byq=collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
# steps: fm_rows_forward_k with TRAIN launches mark steps
fw=[r for r in rows if "fm_rows_forward_k" in r["Kernel_Name"] and ", true" in r["Kernel_Name"]]
print("training forward launches:", len(fw))
if len(fw) > 22:
    w0=int(fw[10]["Start_Timestamp"]); w1=int(fw[20]["Start_Timestamp"])
    print("10 steps span %.1f us -> %.1f us per step" % ((w1-w0)/1e3, (w1-w0)/1e4))
    for q,rs in sorted(byq.items()):
        sel=[r for r in rs if w0 <= int(r["Start_Timestamp"]) < w1]
        if not sel: continue
        busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in sel)
        gaps=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(sel,sel[1:])]
        pos=sorted(g for g in gaps if g>0)
        print("queue",q,"kernels per step %.1f"%(len(sel)/10),"busy %.1f us/step"%(busy/1e4),"idle between its kernels %.1f us/step"%(sum(pos)/1e4),"median gap %.2f us"%(pos[len(pos)//2]/1e3 if pos else 0),"gaps > 20 us per step: %.1f"%(len([g for g in pos if g>20000])/10))
        names=collections.Counter(); cnts=collections.Counter()
        for r in sel:
            nm=r["Kernel_Name"].split("(")[0][-44:]
            names[nm] += int(r["End_Timestamp"])-int(r["Start_Timestamp"]); cnts[nm] += 1
        for n,t in names.most_common(40): print("     %-46s %6.1f us/step in %4.1f launches/step" % (n, t/1e4, cnts[n]/10))
