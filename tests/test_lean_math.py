"""CPU: accuracy of the lean float64 log / exp2 / sincospi used by the kernels (csrc/lean_math.h), measured on the host against a
long double reference over 2e6 samples per argument domain.  The header is host+device, so this is the same
source the kernel compiles (the device build contracts a*b+c into FMAs, which can only tighten the error).
The rsqrt row exercises the host stand-in only (the device calls ocml rsqrt); it is reported, not asserted --
the device function is judged by the GPU parity tests (float64 outputs within 1e-9 of the oracle)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lean_math_accuracy(tmp_path):
    exe = os.path.join(tmp_path, "lean_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "lean_math_check.cpp")])
    rows = {}
    for line in subprocess.check_output([exe], text=True).splitlines():
        name, n, worst, mean = line.split()
        rows[name] = (int(n), float(worst), float(mean))   # (u32div row: total, mismatches, 0)
    for name in ("log_one_minus_u", "log_sinr_ratio", "log_near_one"):
        n, worst, mean = rows[name]
        assert n == 2000000
        assert worst < 1.0, "%s: max error %.3f ulp" % (name, worst)      # fdlibm's bound for this algorithm
        assert mean < 0.3
    n, worst, mean = rows["logc_sinr_ratio"]          # same algorithm, coefficients from the pinned block
    assert worst < 1.0
    for name in ("exp2_fading", "exp2_wide"):          # degree-13 Taylor of 2^r on |r| <= 1/2
        n, worst, mean = rows[name]
        assert worst < 1.0 and mean < 0.3, "%s: %.3f ulp" % (name, worst)
    for name in ("sinpi_0_2", "cospi_0_2"):            # relative error, including next to the zeros
        n, worst, mean = rows[name]
        assert worst < 2.0 and mean < 0.4, "%s: %.3f ulp" % (name, worst)
    n, worst, mean = rows["lm_div"]                    # rcp + Newton division on the two operand ranges the kernels feed it;
    assert n == 2000000 and worst <= 1.0 and mean < 0.3   # host seed is 24 bits, the device's v_rcp_f64 is better
    assert "rsqrt_dist2" in rows
    total, bad, _ = rows["u53_mismatches"]              # csrc/philox.h: fma form of the 53-bit uniform == integer form, bit for bit
    assert total == 8000121 and bad == 0
    total, bad, _ = rows["lane_div_mismatches"]        # csrc/intdiv.h: lane / U, the slot of a lane in the packed kernel
    assert total == 4096 and bad == 0
    total, bad, _ = rows["u32div_mismatches"]          # csrc/intdiv.h: exact a / d for d = 2..9 (action digits)
    assert total == 8 * 9000000 and bad == 0
