for a in "sgd 16 iid" "sgd 16 stratified" "sgd 32 iid" "sgd 64 iid" "sgd_l1 16 iid" "sgd_l1 32 iid" "sgd 16 iid 39"; do
  timeout -k 10 200 python profiles/probes/seq_reassoc_rate.py $a 2>&1 | grep "reassociate=1\|max" | tr '\n' ' '; echo
done
