"""The profile tooling the measured claims rest on (CPU): tools/roofline_table.py joins a rocprofv3 kernel trace with the library's launch
manifest by (kernel instantiation, workgroups), averages per key, marks overlapping dispatches and never prints a fraction above 1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRACE_HEAD = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp",'
              '"End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y",'
              '"Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n')


def _row(i, name, start, end, wgs):
    return f'"KERNEL_DISPATCH","Agent 2",1,0,1,{i},1,"{name}",{i},{start},{end},0,0,64,0,32,256,1,1,{256 * wgs},1,1\n'


def _run(tmp_path, rows, manifest):
    d = tmp_path / "visit"
    (d / "prof").mkdir(parents=True)
    (d / "prof" / "kt_kernel_trace.csv").write_text(TRACE_HEAD + "".join(rows))
    (d / "manifest.tsv").write_text(manifest)
    (d / "commit.txt").write_text("abc1234\n")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), str(d), "--title", "unit test"],
                          capture_output=True, text=True)


def test_roofline_table_joins_by_name_and_workgroups(tmp_path):
    k = "void amp::conv_f16x3_kernel<3, 4, 1, 4, 64>(amp::ConvArgs)"
    rows = [_row(1, k, 0, 100_000, 512), _row(2, k, 200_000, 300_000, 512),          # 100 us each, 512 workgroups: layer A
            _row(3, k, 400_000, 450_000, 128),                                        # 50 us, 128 workgroups: ANOTHER layer on the same instantiation
            _row(4, "void amp::act1d_kernel<16>(float const*, float*)", 500_000, 600_000, 4096)]
    man = ("conv_f16x3_kernel<3, 4, 1, 4, 64>\t512\t41.943\t100.0\tconv 128->128 k=3\n" * 2
           + "conv_f16x3_kernel<3, 4, 1, 4, 64>\t128\t10.0\t20.0\tconv 64->64 k=3\n"
           + "act1d_kernel<16>\t4096\t0.0\t400.0\tActivation1d C=128\n")
    r = _run(tmp_path, rows, man)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "tree abc1234" in out
    lines = [l for l in out.splitlines() if l.startswith("conv_f16x3_kernel")]
    assert len(lines) == 2                                             # two keys, not one guessed shape
    a = next(l for l in lines if " 512 " in l)
    assert "conv 128->128 k=3" in a and " 0.500 " in a                 # 41.943 GFLOP / 100 us = 419.4 TFLOP/s = 0.500 of 838.9
    act = next(l for l in out.splitlines() if l.startswith("act1d_kernel<16>"))
    assert "0.500" in act                                              # 400 MB / 100 us = 4 000 GB/s of 8 000


def test_roofline_table_never_prints_a_fraction_above_one_and_marks_overlap(tmp_path):
    k = "void amp::rb_f16x3_kernel<3, 1, 4, 4, 0, 32>(amp::RbArgs)"
    rows = [_row(1, k, 0, 100_000, 1000), _row(2, k, 50_000, 150_000, 1000)]         # two dispatches sharing half their duration
    man = "rb_f16x3_kernel<3, 1, 4, 4, 0, 32>\t1000\t500.0\t10.0\twhole ResBlock (a manifest that overstates the work)\n" * 2
    r = _run(tmp_path, rows, man)
    assert r.returncode == 1                                           # an inconsistent fraction is an error, not a number
    line = next(l for l in r.stdout.splitlines() if l.startswith("rb_f16x3_kernel"))
    assert "incons" in line and "100%" in line                         # 5 000 TFLOP/s withheld; both dispatches overlap another
    assert "withheld" in r.stdout


def test_isa_mix_counts_instruction_classes(tmp_path):
    s = ("_Z1kv:                                 ; @_Z1kv\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]\n\tv_fma_f32 v0, v1, v2, v3\n\tv_sin_f32_e32 v0, v0\n"
         "\tv_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], 0\n\tds_read_b128 v[0:3], v4\n\tglobal_load_dword v0, v[1:2], off\n\ts_nop 1\n.Lfunc_end0:\n")
    p = tmp_path / "k.s"
    p.write_text(s)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), str(p), "--serial"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "ONE AT A TIME: 0 of 1" in r.stdout                         # the one load is never waited for with vmcnt(0)
    assert "'pk_f32': 1" in r.stdout and "'mfma': 1" in r.stdout and "'trans': 1" in r.stdout and "'lds': 1" in r.stdout and "'vmem': 1" in r.stdout


def test_bench_line_is_compact_and_carries_every_config_at_both_ends():
    """bench.py prints the compact form of its record (the driver keeps the head and the last ~2 KB of the line): <= 6 KB, the
    contract's keys, `roofline`, `cpu_baseline`, and the summary of every BASELINE config as the FIRST and the LAST key -- checked on round 5's
    full record (profiles/r5_bench.json, 12 KB)."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_compact", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.loads(open(os.path.join(root, "profiles", "r5_bench.json")).read().strip().split("\n")[-1])
    line = bench.compact_line(full, os.path.join(root, "gpurun_out", "bench_detail.json"))
    text = json.dumps(line)
    assert len(text) <= 6 * 1024, len(text)
    keys = list(line)
    assert keys[0] == "summary_ms" and keys[-1] == "summary_ms_tail" and line["summary_ms"] == line["summary_ms_tail"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert line[k] == full[k], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) <= 1e-4 * full["roofline"]["frac"]
    assert line["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"] and line["cpu_baseline"]["kind"] == "port"
    tail = text[-2048:]
    for k in ("c3_bigvgan_ms", "c2_strict_fp32_ms", "c5_vits_decode_ms", "c1_clips_ms", "cpu_b4_x_realtime"):
        assert k in tail, k
