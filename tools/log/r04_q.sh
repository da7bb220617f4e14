cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in abl1 abl2; do
  SPECINV_STAMP_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_$w.so python3 tools/obj_stamps.py 2>&1 | grep -v "^$" | tail -4
done
