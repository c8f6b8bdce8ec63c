"""Host test of tools/pmc_traffic.py (the reduction of rocprofv3 --pmc CSVs that bench.py's `roofline.traffic` and
`roofline_e2e.pmc_bytes_per_step` read): kernel-symbol normalisation (demangled AND Itanium-mangled names with bf16 template
arguments) and the segmentation of a counter pass into training steps / forwards."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
pt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pt)


def test_kernel_symbol_normalisation():
    assert pt._sym("void (anonymous namespace)::bn_act_fwd_kernel<0, float, float>(float const*, int)") == "bn_act_fwd_kernel<0,float,float>"
    assert pt._sym("(anonymous namespace)::mel_fwd_kernel(float const*, int)") == "mel_fwd_kernel"
    # mangled form rocprofv3 prints for instantiations with __bf16 arguments
    assert pt._sym("_ZN12_GLOBAL__N_119pw_conv_bf16_kernelILi8ELi1ELi2ELb0EDF16bfEEvPKT3_PKDF16b") == "pw_conv_bf16_kernel<8,1,2,false,bf16,float>"
    assert pt._sym("_ZN12_GLOBAL__N_117bn_act_fwd_kernelILi2EDF16bDF16bEEvPKT0_") == "bn_act_fwd_kernel<2,bf16,bf16>"
    assert pt._sym("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float> >(int)") is None


def _write(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for i, (name, grid, val) in enumerate(rows):
            w.writerow(dict(Dispatch_Id=i + 1, Kernel_Name=name, Grid_Size=grid, Counter_Name=counter, Counter_Value=val))


def test_step_segmentation_and_calibration(tmp_path):
    an = "void (anonymous namespace)::"
    mel, conv, wg = an + "mel_fwd_kernel(float const*)", an + "pw_conv_kernel<4, true, false>(float const*)", an + "pw_wgrad_kernel(float const*)"
    torch_k = "void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float> >(int)"
    calib = [(an + f"calib_copy_kernel<{m}>(float const*, float*, long long)", 1 << 20, 524288.0) for m in range(4)]
    probe = [(mel, 64, 1.0), (conv, 64, 2.0)]                                   # a 4-clip parity probe: smaller grids, dropped
    fwd = [(mel, 4096, 10.0), (conv, 8192, 100.0), (conv, 8192, 60.0)]
    train = [(mel, 4096, 10.0), (conv, 8192, 100.0), (torch_k, 128, 7.0), (wg, 2048, 300.0), (conv, 8192, 50.0)]
    rows = calib + probe + fwd + train + train + fwd
    f_csv, w_csv, out = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "o.json"
    _write(f_csv, "FETCH_SIZE", rows)
    _write(w_csv, "WRITE_SIZE", [(n, g, v * 2 if "calib" in n else v / 2) for n, g, v in rows])
    pt.main(str(f_csv), str(w_csv), str(out))
    doc = json.load(open(out))
    tr, fw = doc["train_step"], doc["forward_step"]
    assert tr["steps_averaged"] == [2, 2] and tr["dispatches_per_step"] == [5, 5]
    assert fw["steps_averaged"] == [2, 2] and fw["dispatches_per_step"] == [3, 3]
    # per-symbol sums over the launches of ONE step; the torch kernel counts in the step total only
    assert tr["kernels"]["pw_conv_kernel<4,true,false>"] == {"launches": 2, "fetch_kib": 150.0, "write_kib": 75.0}
    assert tr["kernels"]["pw_wgrad_kernel"]["fetch_kib"] == 300.0
    assert tr["fetch_kib_total"] == 10.0 + 100.0 + 7.0 + 300.0 + 50.0
    assert fw["kernels"]["pw_conv_kernel<4,true,false>"]["fetch_kib"] == 160.0
    # calibration: 1 GiB known / (counter KiB * 1024): FETCH reads half -> factor 2, WRITE exact -> factor 1
    assert abs(doc["calibration"]["fetch_factor"]["b16"] - 2.0) < 1e-12 and abs(doc["calibration"]["write_factor"]["b16"] - 1.0) < 1e-12
