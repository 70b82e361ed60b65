"""Per-(kernel, grid, workgroup) duration summary of a rocprofv3 --kernel-trace csv: count, median, min (us)."""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else '.'
    files = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
    if not files:
        sys.exit('no kernel_trace.csv under ' + root)
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        key = (r['Kernel_Name'][:70], r['Grid_Size_X'], r['Workgroup_Size_X'])
        d[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items():
        v.sort()
        print(f'{k[0]:70s} grid {k[1]:>8s} wg {k[2]:>5s} n={len(v):4d} med {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f} us')


if __name__ == '__main__':
    main()
