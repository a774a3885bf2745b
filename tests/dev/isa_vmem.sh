# dev aid: compile one .hip file with --save-temps and list the vector-memory instructions, barriers and vmcnt waits of the kernels whose mangled name matches $2
# usage: bash tests/dev/isa_vmem.sh femus_amd/csrc/fh_assemble.hip 'k_cluster_q2hex_sfILi0ELb0ELb1ELb1E'
SRC=$(realpath $1); PAT=$2
mkdir -p /tmp/st && cd /tmp/st && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -x hip -c $SRC -o /tmp/st/a.o --save-temps -Wno-unused-value 2>/dev/null
python3 - "$PAT" <<'PY'
import re, sys, glob
s = open(glob.glob('/tmp/st/*gfx950.s')[0]).read()
md = s[s.index('amdhsa.kernels'):]
for blk in md.split('  - .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if not re.search(sys.argv[1], name): continue
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
    print(name[:70], 'vgpr', g('vgpr_count'), 'spill', g('vgpr_spill_count'), 'sgpr', g('sgpr_count'), 'sspill', g('sgpr_spill_count'))
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    for k, l in enumerate(s[i:j].split('\n')):
        if re.search(r'global_load|flat_load|global_store|flat_store|s_barrier|vmcnt|scratch_', l): print('  ', k, l.strip())
PY
