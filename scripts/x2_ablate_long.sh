# Timing-only ablations of the split score kernel on the long-catalog shape (512 users x 12.5 M items); variants: make -C recboard_amd/csrc var V=...
cd /tmp && export TMPDIR=/tmp
for v in BASE NOMFMA NOAPPEND NOBARRIER; do
  O=/tmp/abl_$v; rm -rf $O; mkdir -p $O
  if [ $v = BASE ]; then L=$GRAFT_REPO_ROOT/recboard_amd/librecengine.so; else L=$GRAFT_REPO_ROOT/recboard_amd/var_$v.so; fi
  RECENGINE_LIB=$L rocprofv3 --kernel-trace -d $O -- python3 $GRAFT_REPO_ROOT/scripts/x2_prof.py 512 12500000 64 > $O/log.txt 2>&1
  f=$(find $O -name "*.db" | head -1)
  echo "$v: $(python3 $GRAFT_REPO_ROOT/scripts/kstats.py $f 20 3 | grep 'score_kernel_reg<64, 28' | cut -c1-100)"
done
