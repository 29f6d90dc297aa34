for ko in 0 1 2 4 7; do
  echo "== FMX_ALLF_KO=$ko"
  FMX_LIB_PATH=profiles/_variants/allf_ko$ko/libfmx.so python profiles/probes/als_iid_levels.py 2>/dev/null | tail -1
done
