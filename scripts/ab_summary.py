"""profiles/r06_handoff_ab.md from the A/B logs of docs/history/scripts/r6_ab.sh (gpurun_out/<tag>/ab.log): one table per session."""
import collections, os, re, statistics, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
out = ['# Round 6: hand-off of the four-wave kernel -- barrier vs flags, wave priorities, cuts (interleaved A/B on one MI355X per session)', '',
       '`docs/history/scripts/r6_ab.sh <tag> "<variants>" <rounds> [bench args]`: every variant is a whole `liboscen_gpu_<tag>.so` (`scripts/build_variant.py`), runs are interleaved',
       'round by round on the same box; `value` = `bench.py --steps 20 --warmup 5` (median region), `first5` = what five cold regions report, kernel ms per 256-frame block',
       'from the in-kernel clock.  Sessions ran on different boxes: compare within a session only.  65 536 voices, fm_voice, unless the tag names a bank size.', '',
       'Variant names: `base` = the library of the tree at that time (barrier, priorities 2,1,1,0 until session r06l; flags 16x2 + 0,1,2,3 from r06m on); `nosync` = no hand-off at all',
       '(results wrong, timing only: the upper bound); `fAxD` / `eAxD` = flags, A-frame chunks, rings of D chunks (f: slot released when the consumer finishes the chunk; e: released once',
       'the chunk\'s values are in registers; `t` = tight poll, no `s_sleep`); `...pr` / `gWXYZ` / `qWXYZ` = flags 16x2 tight with `s_setprio` W,X,Y,Z for the four stages (`h`: 8x4);',
       '`bpr`, `b0123` = the BARRIER with priorities 0,1,1,2 / 0,1,2,3; `d*` = dynamic priority (a wave that had to poll at its chunk top yields) -- rejected; `p*` = progress words',
       'read a chunk ahead -- no gain, removed; `oldtpt*` = round 5\'s TPT update path (not a clean A/B: see ROUND6.md); `flat` = branch-free TPT update on every frame; `cABC` = pipeline',
       'cut (A,B,C) under flags; `n2` / `n4` = the 8-frame shapes with flags, rings of 2 / 4 (`OGC_NARROW_FD`); `prevhost` = the host library before the ramp-end launch.', '']
for tag in sorted(os.listdir(root)):
    f = os.path.join(root, tag, 'ab.log')
    if not (tag.startswith('r06') or tag.startswith('r07')) or not os.path.exists(f):
        continue
    rows = collections.OrderedDict()
    for l in open(f):
        m = re.match(r'(\S+) round (\d+) value (\S+) first5 (\S+) kernel_ms/block (\S+) (\S+)', l)
        if m:
            rows.setdefault(m.group(1), []).append((float(m.group(3)), float(m.group(4)), float(m.group(5)), m.group(6)))
    if not rows:
        continue
    scen = ('--variant survey2 (moving cutoff, feedback, route)' if '_s2' in tag else
            '--steps 188 --warmup 8 (the one-second plan as is)' if 'k188' in tag else
            ('--voices-per-gpu ' + tag.split('_')[-1]) if re.search(r'_\d{5,}$', tag) else "the driver's command")
    out += ['## %s: %s' % (tag, scen), '', '| variant | runs | value (mean) | first5 (mean) | kernel ms/block (mean) | kernel |', '|---|---|---|---|---|---|']
    for k, v in rows.items():
        out.append('| %s | %d | %.3e | %.3e | %.5f | %s |' % (k, len(v), statistics.mean(x[0] for x in v), statistics.mean(x[1] for x in v),
                                                          statistics.mean(x[2] for x in v), v[0][3].split('_')[1]))
    out.append('')
open(os.path.join(os.path.dirname(root), 'profiles', 'r06_handoff_ab.md'), 'w').write('\n'.join(out))
print(len(out), "lines")
