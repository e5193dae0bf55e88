cd $GRAFT_REPO_ROOT
python tools/vit_time.py
for v in NO_TRACEBACK NO_FORWARD NO_DEPUNCT; do
  mkdir -p /tmp/obj_$v
  make -s -C sdrplusplus-dab-radio-plugin_amd/csrc OBJDIR=/tmp/obj_$v OUT=/tmp/lib_$v.so EXTRA=-DDAB_EXP_$v 2>&1 | grep -E "error" 
  DABGPU_LIB=/tmp/lib_$v.so python tools/vit_time.py
done
DABGPU_VITERBI_V0=1 python tools/vit_time.py
