"""Turn the rocprofv3 CSVs of tools/collect_profile.sh into the committed summary
under profiles/: per-kernel time table (kernel-trace --stats) and per-launch HBM traffic
of the dominant kernel from the PMC passes.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts exactly half of a wide coalesced streaming read, so
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.

    python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/r01_<tag>
"""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    hits = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return hits[0] if hits else None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    lines = []
    st = find(os.path.join(src, 'trace'), '*kernel_stats.csv')
    if st:
        rows = list(csv.DictReader(open(st)))
        lines.append('| kernel | calls | total ms | avg us | % |')
        lines.append('|---|---|---|---|---|')
        for r in rows[:14]:
            lines.append('| `%s` | %s | %.3f | %.2f | %s |' % (r['Name'][:70], r['Calls'],
                         float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
        out['kernel_stats'] = [{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage')}
                               for r in rows[:14]]
    # dominant kernel launches in the trace: the ConvLSTM K=36000 convs are the longest conv_igemm<.,2,2> launches
    tr = find(os.path.join(src, 'trace'), '*kernel_trace.csv')
    if tr:
        rows = [r for r in csv.DictReader(open(tr)) if 'conv_igemm' in r['Kernel_Name'] or 'conv_clip' in r['Kernel_Name'] or 'wino_gemm' in r['Kernel_Name']]
        by_grid = {}
        for r in rows:
            key = (r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            by_grid.setdefault(key, []).append(dur)
        top = sorted(by_grid.items(), key=lambda kv: -sum(kv[1]))[:8]
        lines.append('')
        lines.append('| conv_igemm launch shape (grid threads x,y,z) | launches | avg us | total ms |')
        lines.append('|---|---|---|---|')
        for key, durs in top:
            lines.append('| %s grid=(%s,%s,%s) | %d | %.1f | %.2f |' % (key[0], key[1], key[2], key[3], len(durs),
                         sum(durs) / len(durs), sum(durs) / 1e3))
        out['conv_igemm_by_grid'] = [{'grid': k[1:], 'n': len(v), 'avg_us': sum(v) / len(v)} for k, v in top]
        # the dominant kernel of bench.py = the K = 36000 ConvLSTM launches (Conv2 / Gates) of conv_clip_kernel;
        # Conv1 (K = 18000) and layer4's conv2 share the kernel and grid and take about half the time, so
        # split the launches of that kernel at 75 % of its longest launch
        # round 5: with enough tiles the ConvLSTM runs in the Winograd domain and the dominant kernel is wino_gemm_kernel (K = 4000
        # channels x 16 positions for Conv2 / Gates, K = 2000 for Conv1)
        wino = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'wino_gemm' in r['Kernel_Name']]
        if wino:
            # two of a cell update's three GEMMs are K = 4000 (Conv2, Gates), one is K = 2000 (Conv1): the longer two thirds
            wino.sort()
            small, big = wino[: len(wino) // 3], wino[len(wino) // 3:]
            lines.append('')
            lines.append('wino_gemm_kernel launches: %d of K = 4000 x 16 positions (ConvLSTM Conv2 / Gates in the Winograd domain, the dominant '
                         'kernel of bench.py): avg %.1f us; %d shorter ones (Conv1, K = 2000): avg %.1f us'
                         % (len(big), sum(big) / len(big), len(small), sum(small) / max(1, len(small))))
            out['dominant_kernel'] = {'name': 'wino_gemm_kernel', 'launches': len(big), 'avg_us': sum(big) / len(big)}
        clip = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'conv_clip' in r['Kernel_Name']]
        if clip and not wino:
            cut = 0.75 * max(clip)
            big = [d for d in clip if d >= cut]
            small = [d for d in clip if d < cut]
            lines.append('')
            lines.append('conv_clip_kernel launches: %d of K = 36000 (ConvLSTM Conv2 / Gates, the dominant kernel of bench.py): '
                         'avg %.1f us; %d shorter ones (Conv1 K = 18000, layer4 conv2): avg %.1f us'
                         % (len(big), sum(big) / len(big), len(small), sum(small) / max(1, len(small))))
            out['dominant_kernel'] = {'name': 'conv_clip_kernel', 'launches': len(big), 'avg_us': sum(big) / len(big)}
    pm = {}
    is_wino = False
    for name in ('fetch', 'write'):
        f = find(os.path.join(src, name), '*counter_collection.csv')
        if not f:
            continue
        rows = list(csv.DictReader(open(f)))
        # the ConvLSTM convolutions: the clip-resident kernel when it ran, else the generic implicit GEMM
        wino = [r for r in rows if 'wino_gemm' in r['Kernel_Name']]
        is_wino = is_wino or bool(wino)
        clip = wino if wino else [r for r in rows if 'conv_clip' in r['Kernel_Name']]
        rows = clip if clip else [r for r in rows if 'conv_igemm' in r['Kernel_Name']]
        by = {}
        for r in rows:
            key = (r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), r.get('Grid_Size_Y', ''),
                   r.get('Grid_Size_Z', ''))
            by.setdefault(key, []).append(float(r['Counter_Value']))
        pm[name] = by
    if 'fetch' in pm and 'write' in pm:
        # largest-traffic grid = the ConvLSTM convolutions.  That grid carries Conv1 (K = 18000) and
        # Conv2 / Gates (K = 36000) launches; the dominant kernel of bench.py is the K = 36000 shape =
        # the launches with the larger FETCH_SIZE (upper two thirds of the sorted values).
        key = max(pm['fetch'], key=lambda k: sum(pm['fetch'][k]))
        fv = sorted(pm['fetch'][key])
        wv = sorted(pm['write'].get(key, [0]))
        big = fv[len(fv) // 3:]
        fe = sum(big) / len(big)
        wr = sum(wv) / max(1, len(wv))
        traffic = (2 * fe + wr) * 1024
        out['dominant_grid'] = key
        out['FETCH_SIZE_KiB_per_launch_K36000'] = fe
        out['FETCH_SIZE_KiB_per_launch_all'] = sum(fv) / len(fv)
        out['WRITE_SIZE_KiB_per_launch'] = wr
        out['conv_igemm_clstm_bytes_per_launch'] = traffic
        lines.append('')
        # the algorithmic bytes depend on the launch shape (M = pixels of the batch, split-K factor): derived from the
        # write counter instead of quoted for one shape - one f32 slab of the 4000-channel output is M * 16 kB
        if is_wino:
            lines.append('wino_gemm_kernel launches (grid %s): FETCH_SIZE %.0f KiB (K = 4000 launches; %.0f KiB over all three), WRITE_SIZE '
                         '%.0f KiB per launch -> HBM traffic (2*FETCH + WRITE)*1024 = %.1f MB per K = 4000 launch (algorithmic: U 524 MB + V 49 MB '
                         'read, M 98 MB written)' % (key, fe, sum(fv) / len(fv), wr, traffic / 1e6))
        else:
          lines.append('ConvLSTM conv launches (grid %s): FETCH_SIZE %.0f KiB (K=36000 launches; %.0f KiB over all three '
                     'convs), WRITE_SIZE %.0f KiB per launch -> HBM traffic (2*FETCH + WRITE)*1024 = %.1f MB per '
                     'K=36000 launch (algorithmic: 295 MB packed weights + the activations once + the f32 split-K slabs = the '
                     'write counter: %.0f MB = %.1f slabs of an M = 1176 launch (18.8 MB each) / of an M = 1536 launch (24.6 MB each: %.1f))'
                     % (key, fe, sum(fv) / len(fv), wr, traffic / 1e6, wr * 1024 / 1e6, wr * 1024 / 18.8e6, wr * 1024 / 24.6e6))
    # per-kernel HBM bytes and GB/s: FETCH_SIZE / WRITE_SIZE averaged per launch and kernel name (counter passes),
    # duration from the kernel-trace pass
    if tr and find(os.path.join(src, 'fetch'), '*counter_collection.csv') and find(os.path.join(src, 'write'), '*counter_collection.csv'):
        def short(n):
            return n.split('(')[0].replace('void ', '')[:56]
        dur = {}
        for r in csv.DictReader(open(tr)):
            dur.setdefault(short(r['Kernel_Name']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        cnt = {'fetch': {}, 'write': {}}
        for name in ('fetch', 'write'):
            for r in csv.DictReader(open(find(os.path.join(src, name), '*counter_collection.csv'))):
                cnt[name].setdefault(short(r['Kernel_Name']), []).append(float(r['Counter_Value']))
        tab = []
        for k, d in dur.items():
            if k in cnt['fetch'] and k in cnt['write'] and not k.startswith('at::') and 'pack' not in k:
                fe = sum(cnt['fetch'][k]) / len(cnt['fetch'][k])
                wr = sum(cnt['write'][k]) / len(cnt['write'][k])
                us = sum(d) / len(d)
                by = (2 * fe + wr) * 1024
                tab.append((sum(d), k, len(d), us, by, by / us / 1e3))
        tab.sort(reverse=True)
        lines.append('')
        lines.append('| kernel (all launches of that name) | launches | avg us | HBM MB / launch (2*FETCH+WRITE) | GB/s | of 8 TB/s |')
        lines.append('|---|---|---|---|---|---|')
        for _, k, n, us, by, gbs in tab[:24]:
            lines.append('| `%s` | %d | %.1f | %.1f | %.0f | %.2f |' % (k, n, us, by / 1e6, gbs, gbs / 8000.0))
        out['per_kernel_hbm'] = [{'kernel': k, 'launches': n, 'avg_us': us, 'bytes_per_launch': by, 'GBps': gbs} for _, k, n, us, by, gbs in tab]
    os.makedirs(os.path.dirname(dst) or '.', exist_ok=True)
    if len(sys.argv) > 3 and 'conv_igemm_clstm_bytes_per_launch' in out:      # the file bench.py reads `traffic` from
        json.dump({'conv_igemm_clstm_bytes_per_launch': out['conv_igemm_clstm_bytes_per_launch'],
                   'kernel': out.get('dominant_kernel', {}).get('name', 'conv_clip_kernel'),
                   'source': os.path.basename(dst) + '.json'}, open(sys.argv[3], 'w'), indent=1)
    json.dump(out, open(dst + '.json', 'w'), indent=1)
    open(dst + '.md', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
