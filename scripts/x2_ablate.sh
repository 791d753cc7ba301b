# Timing-only ablations of the split score kernel on the bench's trained state (rocprofv3 kernel trace of scripts/x2_prof.py per
# build).  Build the variants first:  for v in NOMFMA NOAPPEND NOBARRIER NOREFILL NOPREFETCH; do make -C recboard_amd/csrc var V=$v; done
# then run this script on the GPU box from the repository root ($GRAFT_REPO_ROOT).
cd /tmp && export TMPDIR=/tmp
X2_DUMP=/tmp/bs.pt python3 $GRAFT_REPO_ROOT/scripts/x2_bench_state.py 220 > /dev/null 2>&1
for v in BASE NOMFMA NOAPPEND NOBARRIER NOREFILL NOPREFETCH; do
  O=$GRAFT_REPO_ROOT/gpurun_out/abl_$v; rm -rf $O; mkdir -p $O
  if [ $v = BASE ]; then L=$GRAFT_REPO_ROOT/recboard_amd/librecengine.so; else L=$GRAFT_REPO_ROOT/recboard_amd/var_$v.so; fi
  RECENGINE_LIB=$L X2_STATE=/tmp/bs.pt rocprofv3 --kernel-trace -d $O -- python3 $GRAFT_REPO_ROOT/scripts/x2_prof.py > $O/log.txt 2>&1
  f=$(find $O -name "*.db" | head -1)
  echo "$v: $(python3 $GRAFT_REPO_ROOT/scripts/kstats.py $f 20 3 | grep 'score_kernel_reg<64, 28')"
done
