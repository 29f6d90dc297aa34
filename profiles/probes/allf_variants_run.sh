# the one-wave feature-major kernel's build-time choices side by side on one box (profiles/variant_build.sh NAME -DFMX_ALLF_AHEAD=a -DFMX_ALLF_WPE=w)
for v in profiles/_variants/*/; do
  echo "== $v"
  FMX_LIB_PATH=${v}libfmx.so python profiles/probes/als_iid_levels.py 2>/dev/null | tail -1
done
