cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { export GNET_EXTRA_FLAGS="$1"; python -m gossipnet_amd.build > /dev/null 2>&1; echo "[$1] $(timeout 300 python tools/kprobe.py 8 | grep -o "pack=[0-9.]*" | tr "\n" " ")"; }
run "-DPACK_X=8"
run "-DPACK_X=32"
run "-DPACK_X=128"
