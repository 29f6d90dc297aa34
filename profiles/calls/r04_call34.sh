#!/bin/bash
# round 4, thirty-fourth GPU call: the PMC passes again with the L2's fabric reads counted by size (pmc_run.sh group 7), SGD and the Gibbs sweep; the default line after them
export TMPDIR=/tmp
O=gpurun_out/regen4
mkdir -p $O
FMX_ROWS_SERIAL=1 bash profiles/pmc_run.sh $O/pmc_sgd --no-extras > $O/pmc_sgd.log 2>&1; echo "pmc sgd rc=$?"
bash profiles/pmc_run.sh $O/pmc_mcmc --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc.log 2>&1; echo "pmc mcmc rc=$?"
bash profiles/pmc_run.sh $O/pmc_criteo --workload criteo --no-extras > $O/pmc_criteo.log 2>&1; echo "pmc criteo rc=$?"
bash profiles/pmc_run.sh $O/pmc_ftrl --solver ftrl --no-extras > $O/pmc_ftrl.log 2>&1; echo "pmc ftrl rc=$?"
for n in sgd mcmc criteo ftrl; do cp $O/pmc_$n/pmc_summary.json gpurun_out/r04_pmc_summary_$n.json; done
rm -rf $O/pmc_sgd/pass* $O/pmc_mcmc/pass* $O/pmc_criteo/pass* $O/pmc_ftrl/pass*
python3 -c "
import json
for n in ('sgd','mcmc','criteo','ftrl'):
    d=json.load(open('gpurun_out/r04_pmc_summary_%s.json'%n))
    for k,v in d.items():
        if isinstance(v,dict) and 'fabric_bytes_per_launch' in v: print(n,k,'fabric %.1f MB'%(v['fabric_bytes_per_launch']/1e6),'128B frac %.3f'%v['fabric_reads_128B_frac'],'x2 bound %.1f MB'%(v['traffic_bytes_per_launch']/1e6))
"
