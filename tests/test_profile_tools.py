"""The two scripts that fold rocprofv3 output into profiles/*.json, on synthetic CSVs (CPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_collect_pmc_and_mfma(tmp_path):
    f = tmp_path / "f.csv"; w = tmp_path / "w.csv"; m = tmp_path / "m.csv"; k = tmp_path / "k.csv"
    hdr = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    f.write_text(hdr + '1,"void ppv::a<1>(int)",FETCH_SIZE,1000\n2,"void ppv::a<1>(int)",FETCH_SIZE,3000\n3,"ppv::b(int)",FETCH_SIZE,10\n')
    w.write_text(hdr + '1,"void ppv::a<1>(int)",WRITE_SIZE,500\n2,"void ppv::a<1>(int)",WRITE_SIZE,500\n')
    m.write_text(hdr + '1,"void ppv::a<1>(int)",SQ_VALU_MFMA_BUSY_CYCLES,2457600\n1,"void ppv::a<1>(int)",SQ_BUSY_CYCLES,64000\n')
    k.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n"void ppv::a<1>(int)",2,20000,10000,100,1,1,0\n')
    out = tmp_path / "o.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_pmc.py"), str(f), str(w), str(out)])
    d = json.load(open(out))
    a = d["per_kernel"]["ppv::a<1>"]
    assert a["launches_2steps"] == 2 and a["fetch_bytes_per_launch_corrected"] == 2 * 4000 * 1024 / 2 and a["write_bytes_per_launch"] == 512000
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_mfma.py"), str(m), str(k), str(out)])
    d = json.load(open(out))["per_kernel"]["ppv::a<1>"]
    assert abs(d["mfma_busy_frac"] - 2457600 / (1024 * 10000 * 2.4)) < 1e-4
    assert abs(d["mfma_busy_frac_at_actual_clock"] - 2457600 / (32 * 64000)) < 1e-4
