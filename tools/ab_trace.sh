# A/B of library builds under the kernel trace of the headline step: LIBS="path1 path2" PAT="chain|wgrad" bash tools/ab_trace.sh
# (per build: the per-kernel averages matching PAT and the length of one traced step; default build = empty string "-")
for lib in $LIBS; do
  echo "==== $lib"
  if [ "$lib" = "-" ]; then unset FLID_TG_LIB; else export FLID_TG_LIB=$GRAFT_REPO_ROOT/$lib; fi
  TAG=ab_$(basename $lib .so) LINES_OUT=60 bash tools/step_prof.sh | grep -E "${PAT:-chain}|one step" | cut -c1-110
done
