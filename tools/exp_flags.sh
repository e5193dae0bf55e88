# usage: tools/exp_flags.sh <timing script> "<flags1>" "<flags2>" ...   (GPU box)
cd $GRAFT_REPO_ROOT
script=$1; shift
python $script
i=0
for f in "$@"; do
  i=$((i+1)); mkdir -p /tmp/objx_$i
  make -s -C sdrplusplus-dab-radio-plugin_amd/csrc OBJDIR=/tmp/objx_$i OUT=/tmp/libx_$i.so EXTRA="$f" 2>&1 | grep -E "error"
  echo "flags: $f"; DABGPU_LIB=/tmp/libx_$i.so python $script
done
