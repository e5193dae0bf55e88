cd $GRAFT_REPO_ROOT
python tools/ofdm_time.py
for v in NOPLL NOCYC NOEPI; do
  mkdir -p /tmp/obj_$v
  make -s -C sdrplusplus-dab-radio-plugin_amd/csrc OBJDIR=/tmp/obj_$v OUT=/tmp/lib_$v.so EXTRA=-DDAB_EXP_$v 2>&1 | grep -E "error"
  DABGPU_LIB=/tmp/lib_$v.so python tools/ofdm_time.py
done
DABGPU_OFDM_GROUP=25 python tools/ofdm_time.py
DABGPU_OFDM_V0=1 python tools/ofdm_time.py
