#!/bin/bash
# round 5: the whole GPU suite, then the PMC passes (profiles/pmc_run.sh: one counter group per pass) for the lines bench.py prints: configs[1] on the i.i.d.
# column law (the headline since round 5), its fp64-state form, configs[2], configs[4] in the level-order form, configs[3] resident
export TMPDIR=/tmp
O=gpurun_out/regen5
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
bash profiles/pmc_run.sh $O/pmc_sgd --columns iid --no-extras > $O/pmc_sgd.log 2>&1; echo "pmc sgd rc=$?"
bash profiles/pmc_run.sh $O/pmc_sgd_fp64 --columns iid --state-fp64 --no-extras > $O/pmc_sgd_fp64.log 2>&1; echo "pmc sgd fp64 rc=$?"
bash profiles/pmc_run.sh $O/pmc_mcmc --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc.log 2>&1; echo "pmc mcmc rc=$?"
bash profiles/pmc_run.sh $O/pmc_ftrl --solver ftrl --columns iid --no-extras > $O/pmc_ftrl.log 2>&1; echo "pmc ftrl rc=$?"
bash profiles/pmc_run.sh $O/pmc_criteo --workload criteo --no-extras > $O/pmc_criteo.log 2>&1; echo "pmc criteo rc=$?"
for n in sgd sgd_fp64 mcmc ftrl criteo; do cp $O/pmc_$n/pmc_summary.json gpurun_out/r05_pmc_summary_$n.json; done
rm -rf $O/pmc_*/pass*
python3 -c "
import json
for n in ('sgd','sgd_fp64','mcmc','ftrl','criteo'):
    d=json.load(open('gpurun_out/r05_pmc_summary_%s.json'%n))
    for k,v in d.items():
        if isinstance(v,dict) and 'traffic_bytes_per_launch' in v: print(n,k,'fabric %.1f MB'%(v.get('fabric_bytes_per_launch',0)/1e6),'128B frac %.3f'%v.get('fabric_reads_128B_frac',0),'x2 bound %.1f MB'%(v['traffic_bytes_per_launch']/1e6), 'L2 hit %.2f' % v.get('l2_hit_rate', -1))
"
