# development aid: instruction mix of the largest loop of a kernel in the device assembly (hipcc --cuda-device-only -S)
import re, collections, sys
path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_ZN\w*%s\w*:' % name, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
loops.sort(reverse=True)
print('function lines', len(body), 'loops', loops[:4])
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, a, b = loops[which]
cnt = collections.Counter()
for l in body[a:b + 1]:
    l = l.strip()
    if not l or l[0] in ';.': continue
    cnt[l.split()[0]] += 1
print('total', sum(cnt.values()), 'valu', sum(v for k, v in cnt.items() if k.startswith('v_')), 'salu', sum(v for k, v in cnt.items() if k.startswith('s_')))
for k, v in cnt.most_common(70): print('%-28s %d' % (k, v))
