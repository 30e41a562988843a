"""What `roofline.frac` of the bench line should read, from the kernel trace of the same command: the projection launches
INSIDE the timed steps are the last 4 x K launches before the 5 stand-alone launches `bench.py` makes after the timed region
(`time_dominant_kernel`); the launches before them belong to the settle seconds and the warm-up steps, where the clocks are
still ramping (they run 3-5 % slower).   python tools/dominant_summary.py profiles/r05_dominant_launches.csv [--steps 10] [--encoders 4]"""
import argparse
import csv
import statistics as st

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--encoders', type=int, default=4)
ap.add_argument('--standalone', type=int, default=5)
a = ap.parse_args()
TF, PEAK = 0.8417, 157.3
for kind, tail in (('NT projection', a.standalone), ('TN weight gradient', 0)):
    d = [float(r['duration_ms']) for r in csv.DictReader(open(a.csv)) if kind in r['kernel'] and r['kernel'].startswith('exact')]
    if not d:
        continue
    n = a.steps * a.encoders
    timed = d[len(d) - tail - n:len(d) - tail]
    print('%-20s %3d launches in the trace: mean %.3f ms = %.4f of peak | the %d inside the %d timed steps: mean %.3f ms = **%.4f** '
          '(median %.3f, min %.3f)%s' % (kind, len(d), st.mean(d), TF / st.mean(d) * 1e3 / PEAK, n, a.steps, st.mean(timed),
                                         TF / st.mean(timed) * 1e3 / PEAK, st.median(timed), min(timed),
                                         ' | the %d stand-alone: mean %.3f ms = %.4f' % (tail, st.mean(d[-tail:]), TF / st.mean(d[-tail:]) * 1e3 / PEAK) if tail else ''))
