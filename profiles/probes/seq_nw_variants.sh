for nw in 0 10 12 14; do
  if [ $nw = 0 ]; then unset FMX_LIB_PATH; else export FMX_LIB_PATH=$PWD/profiles/_variants/seqnw$nw/libfmx.so; fi
  echo "NW=$nw (0 = the product's 8 waves): $(timeout -k 10 200 python profiles/seq_phase_ticks.py 2>&1 | tail -1)"
done
FMX_LIB_PATH=$PWD/profiles/_variants/seqnw12/libfmx.so timeout -k 10 300 python -m pytest tests/test_gpu_seq_window.py -x -q 2>&1 | tail -2
