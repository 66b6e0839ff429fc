"""profiles/roofline_traffic.json rows from the PMC summaries of tools/profile_all.sh (tools/pmc_summary.py output),
then tools/traffic_hashes.py --update.  Usage: python tools/update_traffic.py <tag> (reads profiles/<tag>_fwd_pmc_summary.txt
and profiles/<tag>_fwdbwd_pmc_summary.txt)."""
import json, re, subprocess, sys
tag = sys.argv[1]
ROWS = {  # row -> (summary, kernel header prefix)
    'rdb_chain': ('fwd', 'rdb_chain<f16,forward>'),
    'upconv_subpix_c64': ('fwd', 'conv<f16,k2,s1,ups3,wr2,wc1,ncg4,ncw2,wlds1,1x10,bwd0>'),
    'conv3x3_c64': ('fwd', 'conv<f16,k3,s1,ups0,wr4,wc1,ncg1,ncw2,wlds1,1x10,bwd0>'),
    'conv3x3_c32': ('fwd', 'conv<f16,k3,s1,ups0,wr4,wc1,ncg1,ncw1,wlds1,1x10,bwd0>'),
    'rdb_chain_train': ('fwdbwd', 'rdb_chain<f16,train-forward>'),
    'rdb_chain_bwd': ('fwdbwd', 'rdb_chain<f16,backward>'),
    'rdb_wgrad': ('fwdbwd', 'rdb_wgrad_kernel'),
}
def parse(path):
    out, cur = {}, None
    for l in open(path):
        m = re.match(r'(\S.*) dispatches \d+', l)
        if m: cur = m.group(1)
        m = re.search(r'HBM-side read ~ ([\d.]+) MB .* write ~ ([\d.]+) MB', l)
        if m and cur: out[cur] = (float(m.group(1)), float(m.group(2)))
    return out
import os
if os.path.exists('profiles/%s_train_pmc_summary.txt' % tag):      # tools/pmc_train.sh: the train step's shape
    ROWS.update({'rdb_chain_train@train': ('train', 'rdb_chain<f16,train-forward>'), 'rdb_chain_bwd@train': ('train', 'rdb_chain<f16,backward>'),
                 'rdb_wgrad@train': ('train', 'rdb_wgrad_follow_kernel')})     # round 6: the follower form of the pass (csrc/rdb_wgrad.hip)
S = {k: parse('profiles/%s_%s_pmc_summary.txt' % (tag, k)) for k in ('fwd', 'fwdbwd', 'train') if os.path.exists('profiles/%s_%s_pmc_summary.txt' % (tag, k))}
J = json.load(open('profiles/roofline_traffic.json'))
for row, (which, name) in ROWS.items():
    r, w = S[which][name]
    J.setdefault(row, {})
    J[row]['read_bytes'], J[row]['write_bytes'] = int(round(r * 1e6)), int(round(w * 1e6))
    J[row]['note'] = 'round %s, profiles/%s_%s_pmc_summary.txt (FETCH_SIZE x2 gfx950 correction / WRITE_SIZE, MB = KB/1024 x 1e6 as the summary prints it)' % (tag[1:].lstrip('0'), tag, which)
    print(row, J[row]['read_bytes'], J[row]['write_bytes'])
J['_source'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, round %s: profiles/%s_fwd_pmc_summary.txt (forward kernels), '
                'profiles/%s_fwdbwd_pmc_summary.txt (training kernels); tools/profile_all.sh %s') % (tag[1:].lstrip('0'), tag, tag, tag)
# the profiled box's average launch duration of the dominant forward kernel (rocprofv3 --kernel-trace --stats of the same
# command): bench.py prints it next to its own HIP-event figure — the two come from DIFFERENT boxes (+-3.5 %)
import csv
st = 'profiles/%s_fwd_kernel_stats.csv' % tag
if os.path.exists(st):
    for r in csv.DictReader(open(st)):
        if 'rdb_chain_kernel' in r['Name']:
            J['rdb_chain']['profiled_avg_launch_us'] = round(float(r['AverageNs']) / 1e3, 2)
            J['rdb_chain']['profiled_min_launch_us'] = round(float(r['MinNs']) / 1e3, 2)
            J['rdb_chain']['profiled_calls'] = int(r['Calls'])
            J['rdb_chain']['profiled_source'] = st
            break
json.dump(J, open('profiles/roofline_traffic.json', 'w'), indent=2)
subprocess.check_call([sys.executable, 'tools/traffic_hashes.py', '--update'])
