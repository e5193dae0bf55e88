cd $GRAFT_REPO_ROOT
python tools/ofdm_time.py
for v in "$@"; do
  mkdir -p /tmp/obj_$v
  flags=$(echo $v | sed 's/+/ -DDAB_EXP_/g; s/^/-DDAB_EXP_/')
  make -s -C sdrplusplus-dab-radio-plugin_amd/csrc OBJDIR=/tmp/obj_$v OUT=/tmp/lib_$v.so EXTRA="$flags" 2>&1 | grep -E "error"
  DABGPU_LIB=/tmp/lib_$v.so python tools/ofdm_time.py
done
