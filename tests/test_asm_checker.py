"""tools/check_asm_loads.py on hand-made listings: it must see a register read while an inline-asm load into it may still
be in flight along SOME path (including round a loop back-edge), and must not flag a block that is only reachable after the
loads were waited for (the shape hipcc emits when it places a tail block between a loop and its exit)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "tools", "check_asm_loads.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def listing(body):
    return "k:\n" + body + "\n.end_amdhsa_kernel\n"


LOAD = ";;#ASMSTART\nglobal_load_dwordx4 v[10:13], v[2:3], off\n;;#ASMEND\n"
WAIT0 = ";;#ASMSTART\ns_waitcnt vmcnt(0)\n;;#ASMEND\n"


def test_straight_line_violation_and_wait(capsys):
    assert chk.check(listing(LOAD + "v_mov_b32_e32 v20, v11\n" + WAIT0 + "s_endpgm"), "k") == 1
    assert chk.check(listing(LOAD + WAIT0 + "v_mov_b32_e32 v20, v11\ns_endpgm"), "k") == 0


def test_vmcnt_counts_younger_loads(capsys):
    two = LOAD + ";;#ASMSTART\nglobal_load_dwordx4 v[14:17], v[2:3], off\n;;#ASMEND\n"
    wait1 = ";;#ASMSTART\ns_waitcnt vmcnt(1)\n;;#ASMEND\n"
    assert chk.check(listing(two + wait1 + "v_mov_b32_e32 v20, v11\n" + WAIT0 + "s_endpgm"), "k") == 0
    assert chk.check(listing(two + wait1 + "v_mov_b32_e32 v20, v15\n" + WAIT0 + "s_endpgm"), "k") == 1


def test_violation_round_a_back_edge(capsys):
    # the read sits BEFORE the load in program order; only the loop's back-edge brings the in-flight state to it
    body = ".LBB0_1:\nv_mov_b32_e32 v20, v11\n" + LOAD + "s_cbranch_scc0 .LBB0_1\n" + WAIT0 + "s_endpgm"
    assert chk.check(listing(body), "k") == 1


def test_block_placed_after_the_loop_is_judged_by_its_predecessors(capsys):
    # .LBB0_2 follows the loop in the listing but is entered only from the top, before any load was issued
    body = ("s_cbranch_scc1 .LBB0_2\n"
            ".LBB0_1:\n" + LOAD + "s_cbranch_scc0 .LBB0_1\ns_branch .LBB0_3\n"
            ".LBB0_2:\nv_mov_b32_e32 v20, v11\ns_endpgm\n"
            ".LBB0_3:\n" + WAIT0 + "v_mov_b32_e32 v21, v12\ns_endpgm")
    assert chk.check(listing(body), "k") == 0
    # the same block entered from the loop exit as well: now it is a violation
    body2 = body.replace("s_branch .LBB0_3", "s_branch .LBB0_2")
    assert chk.check(listing(body2), "k") == 1
