cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in ${WAVES:-4}; do
  SPECINV_STAMP_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_stamps$w.so python3 tools/obj_stamps.py 2>&1 | grep -v "^$" | tail -15
done
