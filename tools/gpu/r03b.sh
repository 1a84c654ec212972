cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $O/r03b_pmc_lds -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/r03b_pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O/r03b_pmc_lds2 -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/r03b_pmc_lds2.log 2>&1
python - <<PY
import csv,collections,glob
for d in ("$O/r03b_pmc_lds","$O/r03b_pmc_lds2"):
    for f in glob.glob(d+"/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            if "k_score" in k or "k_conj" in k:
                agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k,v in agg.items(): print(k, {c:"%.4g"%(x/len(n[k])) for c,x in v.items()})
PY
tail -3 $O/r03b_pmc_lds.log
