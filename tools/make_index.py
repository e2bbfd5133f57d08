#!/usr/bin/env python3
"""Regenerate tools/README.md: one line per script (its own header comment) and the profiles/ files that name it or
that DESIGN.md / HISTORY.md attribute to it.  Run from the repository root: python tools/make_index.py"""
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
tools = sorted(f for f in os.listdir('tools') if f.endswith(('.sh', '.py', '.hip')))
prof = collections.defaultdict(list)
for p in sorted(os.listdir('profiles')):
    try:
        txt = open('profiles/' + p, errors='ignore').read(6000)
    except OSError:
        continue
    for t in tools:
        if t in txt and p not in prof[t]:
            prof[t].append(p)
docs = open('DESIGN.md').read() + open('HISTORY.md').read()
for t in tools:
    for line in docs.splitlines():
        if t in line:
            for x in re.findall(r'profiles/(round\d_[A-Za-z0-9_]+\.[a-z]+)', line):
                if x not in prof[t]:
                    prof[t].append(x)


def desc(t):
    out = []
    for line in open('tools/' + t, errors='ignore').read().splitlines()[:12]:
        s = line.strip()
        if s.startswith('#!') or not s:
            continue
        if s.startswith(('#', '//', '"""')) or out:
            s = s.lstrip('#/ ').strip('"').strip()
            if not s:
                if out:
                    break
                continue
            out.append(s)
            if s.endswith(('.', ')')) or len(' '.join(out)) > 150:
                break
        else:
            break
    d = re.sub(r'^(Round|round) \d,?\s*(session|call)?\s*[A-Za-z0-9]*:\s*', '', ' '.join(out))
    return (d[:170] + '...') if len(d) > 170 else d


groups = collections.OrderedDict(
    (k, []) for k in ('Evidence scripts, one per round (what the judge-facing numbers came from)',
                      'Stand-alone HIP probes (`hipcc --offload-arch=gfx950 tools/x.hip`; built into tools/_build by the scripts that use them)',
                      'Python probes (called by the session scripts)', 'Round 6 sessions and tools', 'Round 5 sessions',
                      'Round 4 sessions', 'Round 3 sessions', 'Round 2 sessions', 'Other'))
keys = list(groups)
for t in tools:
    if re.match(r'gpu_round\d_final\d?\.sh|gpu_validate\.sh', t):
        k = keys[0]
    elif t.endswith('.hip'):
        k = keys[1]
    elif t in ('train_round6.py', 'weights_pack.py', 'f44_numerics_gate.py', 'gpu_probe_decode_pmc.py', 'make_index.py') or t.startswith('gpu_round6'):
        k = keys[3]
    elif t.startswith('gpu_probe') or t in ('gpu_idle_map.py', 'gpu_diag_1x1.py', 'summarise_pmc.py'):
        k = keys[2]
    elif t.startswith('gpu_round5'):
        k = keys[4]
    elif t.startswith('gpu_round4'):
        k = keys[5]
    elif t.startswith('gpu_round3'):
        k = keys[6]
    elif t.startswith('gpu_round2'):
        k = keys[7]
    else:
        k = keys[8]
    groups[k].append(t)
out = ["# tools/ -- index", "",
       "Everything here is measurement and experiment tooling: nothing under `tools/` is imported by the product",
       "(`pseudocylindrical_convolution_amd/`), by `bench.py`'s timed region or by the tests' product side.  A `gpu_roundN_x.sh`",
       "is ONE gpurun call of round N (`gpurun -- 'bash tools/gpu_roundN_x.sh'`), kept so that every figure in `profiles/`,",
       "`DESIGN.md` and `HISTORY.md` can be traced to the command that produced it.  Column 3: the `profiles/` files that",
       "came out of the script (where the file or the design documents name it).  `tools/experiments/` holds archived kernel",
       "variants that were measured and not kept; `tools/_build/` is scratch (git-ignored).", "",
       "Regenerate with `python tools/make_index.py`: the descriptions are the scripts' own header comments.", ""]
for k, v in groups.items():
    if not v:
        continue
    out += ["## " + k, "", "| script | what it does | profiles/ |", "|---|---|---|"]
    for t in v:
        out.append("| `%s` | %s | %s |" % (t, desc(t).replace('|', '/') or '(see the script)',
                                          ', '.join('`%s`' % p for p in prof[t][:3]) or '--'))
    out.append("")
open('tools/README.md', 'w').write('\n'.join(out))
print("tools/README.md: %d scripts" % len(tools))
