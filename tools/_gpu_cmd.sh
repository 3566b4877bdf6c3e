rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1
python3 -c "from clonealign_amd import engine as E; print('ca_build_id', E.build_id())"
echo "sources: $(cat .git_rev 2>/dev/null)"
python3 -m pytest tests -x -q -m gpu -rs 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -16
echo "== __graft_entry__.smoke()"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
