"""Print the kernel launch sequence of one train step from a rocprofv3 kernel trace (steps end at adamw_kernel).
usage: launch_sequence.py <trace dir> [step index] [--torch-only]"""
import csv
import glob
import re
import sys

TAGS = ('FillFunctor', 'CUDAFunctor_add<float>', 'CUDAFunctor_add<long>', 'direct_copy', 'MulFunctor', 'CompareFunctor',
        'CatArrayBatchedCopy', 'index_elementwise', 'random_from_to', 'uniform_and_transform', 'MeanOps', 'write_indices',
        'bernoulli', 'masked', 'where', 'nonzero', 'LpNorm', 'BinaryOpScalar', 'DivFunctor', 'copyBuffer', 'fillBuffer',
        'sigmoid', 'index_copy', 'indexSelect', 'floor', 'rsub', 'sum')


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    m = re.match(r'([A-Za-z_0-9:]+)', n)
    s = m.group(1) if m else n[:30]
    for key in TAGS:
        if key in n:
            s += '[' + key + ']'
    return s[-70:]


def main():
    f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))
    step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
    a, b = ends[step] + 1, ends[step + 1] + 1
    seq = [(short(r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[a:b]]
    torch_only = '--torch-only' in sys.argv
    print(len(seq), 'launches in step')
    out, prev, cnt, us = [], None, 0, 0.0
    for s, t in seq + [(None, 0)]:
        if s == prev:
            cnt += 1
            us += t
            continue
        if prev is not None:
            out.append(f"{prev} x{cnt}  {us:.0f} us")
        prev, cnt, us = s, 1, t
    for line in out:
        if not torch_only or '[' in line or 'elementwise' in line or 'rocprim' in line or 'reduce' in line:
            print(line)


main()
