"""-m "not gpu": static guard over the device code (tools/isa_guard.py).  Round 5's box-dependent wrong answer was
a packed fp32 multiply whose LOW result takes the HIGH register of a source pair (`v_pk_mul_f32 ... op_sel:[0,1]`), emitted by
hipcc's SLP vectoriser, losing its low product whenever a second wave shared the SIMD.  No kernel that can share a SIMD may
contain a packed-fp32 instruction with a set op_sel bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_scanner_recognises_the_instruction_form(tmp_path):
    import isa_guard
    asm = tmp_path / 'x.s'
    asm.write_text('\n'.join([
        'kernel_a:', '\tv_pk_mul_f32 v[2:3], v[82:83], v[8:9] op_sel:[0,1]', '\tv_pk_add_f32 v[0:1], v[0:1], v[2:3]',
        '\tv_pk_fma_f32 v[4:5], v[4:5], v[6:7], s[2:3] op_sel_hi:[1,1,0]', '\ts_endpgm', '; Occupancy: 2',
        'kernel_b:', '\tv_pk_mul_f32 v[8:9], s[8:9], v[6:7] op_sel:[1,0]', '\tv_pk_mul_f16 v1, v2, v3 op_sel:[0,1]', '; Occupancy: 1',
        'kernel_c:', '\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]', '; Occupancy: 8', '']))
    res = isa_guard.scan(str(asm))
    assert [len(res[k][0]) for k in ('kernel_a', 'kernel_b', 'kernel_c')] == [1, 1, 0]      # op_sel_hi forms and f16 do not count
    assert [res[k][1] for k in ('kernel_a', 'kernel_b', 'kernel_c')] == [2, 1, 8]


def test_no_cross_selected_packed_fp32_in_kernels_that_share_a_simd():
    """every .hip of the library compiled to device assembly with the library's own flags (26 s on 8 cores)"""
    import pytest
    import isa_guard
    from rsprompter_amd import build
    if not os.path.exists(build.HIPCC):
        pytest.skip('hipcc is not installed here: nothing to compile the kernels with')
    table, viol = isa_guard.run()
    for row in table:
        print(row)
    assert not viol, viol
    # ... and since the fused upscaler's dot is pinned like sam_upscale2_kernel's sums, in no kernel at all (upscale.hip and
    # t2i_fold.hip keep hipcc's SLP packing, -fslp-vectorize on their first line: neither gets the form from it)
    assert table == [], table
