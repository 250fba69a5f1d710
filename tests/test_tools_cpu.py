"""tools/trace_wall.py: the wall-time attribution the second half of round 5 worked from, on a trace small enough to
check by hand."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wall_time_attribution_accounts_for_overlap(tmp_path):
    # three kernels on two queues, times in ns: A [0, 10 ms) and B [5, 15 ms) overlap for 5 ms, idle [15, 20), C [20, 30) alone
    trace = tmp_path / "kernel_trace.csv"
    trace.write_text("Kernel_Name,Queue_Id,Start_Timestamp,End_Timestamp\n"
                     "A,1,0,10000000\nB,2,5000000,15000000\nC,1,20000000,30000000\n")
    line = tmp_path / "bench_line.txt"
    line.write_text('{"steps": 1, "ms_per_step": 30.0}\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_wall.py"), str(trace), str(line)],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].startswith("# timed region 1 steps x 30.000 ms; idle 5.000 ms/step; alone 20.000; shared 5.000")
    rows = {}
    for l in out[2:]:
        if l.startswith("#"):
            continue
        name, calls, total, alone, shared, wall = l.replace('"', "").split(",")
        rows[name] = (float(calls), float(total), float(alone), float(shared), float(wall))
    assert rows["A"] == (1.0, 10.0, 5.0, 2.5, 7.5)
    assert rows["B"] == (1.0, 10.0, 5.0, 2.5, 7.5)
    assert rows["C"] == (1.0, 10.0, 10.0, 0.0, 10.0)
    queues = [l for l in out if l.startswith("# queue")]
    assert queues == ["# queue 1  15.000  2.500", "# queue 2  5.000  2.500"]
