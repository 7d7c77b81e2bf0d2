"""A few seeds of tools/fuzz_parity.py: random small jobs (shapes, k lists, filter sizes,
groupings, thresholds, N / lower-case / IUPAC drafts) through both stage-A schedules, whole path,
against the oracle's batch driver."""
import pathlib
import sys

import pytest

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tools"))
pytestmark = pytest.mark.gpu


def test_random_jobs_match_the_oracle(oracle, monkeypatch):
    import fuzz_parity
    from goldpolish_amd import core
    for name in ("GP_STAGE_A_LOCAL", "GP_STAGE_A_CENSUS", "GP_STAGE_A_CENSUS_MIN", "GP_STAGE_A_LEVELS", "GP_STAGE_A_LEVELS_LOAD"):
        monkeypatch.setenv(name, "0")   # run_case sets them per schedule; put back when the test ends
    with core.Context(0) as ctx:
        # 1086: neighbouring gaps closer than two flanks, both closed (overlapping fills);
        # 20621, 20645: near-saturated filters make ntEdit insert until a contig outgrows its slot
        # (gp_job_polish grows the slots and runs the cascade again); 300029: nearly empty filters,
        # low ratios: a 9 kbp contig leaves the cascade at 42 kbp (the walker's pass count
        # follows the edited length, not the input's)
        # 400000+: several contigs per batch with different thresholds on small filters
        for seed in [1086, 20621, 20645, 300029] + list(range(7000, 7009)) + list(range(400000, 400012)):
            for mode in ("0", "1", "2", "3"):   # grid-wide, cbf_local plain, cbf_local behind the census, the census level by level
                assert fuzz_parity.run_case(core, ctx, oracle, seed, mode) is None, (seed, mode)


def test_random_jobs_with_contigs_walked_as_segments(oracle, monkeypatch):
    """the same with every contig of 4 kbp and more cut into segments of 2048 k-mer starts (and, for
    some seeds, at the nominal positions whatever the records say): tiny k on tiny filters makes
    ntEdit refuse insertions, drop nodes and end contigs early -- what the segment walk has to hand
    back to the one-workgroup walk.  910080: a refused insertion's dead nodes end the contig's OUTPUT
    (the reference's writer stops at the first one); segments that did not know gave 18 994 bases
    for the oracle's 11 816."""
    import fuzz_parity
    from goldpolish_amd import core
    for name in ("GP_STAGE_A_LOCAL", "GP_STAGE_A_CENSUS", "GP_STAGE_A_CENSUS_MIN"):
        monkeypatch.setenv(name, "0")
    monkeypatch.setenv("GP_WALK_SEG", "2048")
    monkeypatch.setenv("GP_WALK_SEG_MIN", "0")
    with core.Context(0) as ctx:
        for seed in [910080, 20621, 20645, 300029] + list(range(910000, 910012)):
            assert fuzz_parity.run_case(core, ctx, oracle, seed, "2") is None, seed
        monkeypatch.setenv("GP_WALK_SEG_FORCE", "1")
        for seed in [910080] + list(range(920000, 920008)):
            assert fuzz_parity.run_case(core, ctx, oracle, seed, "2") is None, ("forced", seed)


def test_random_jobs_with_128_thread_walkers(oracle, monkeypatch):
    """the walk as jobs of many contigs run it: four waves per SIMD, 128-thread workgroups (a wave
    per substitute takes two turns, every loop strides by the workgroup's real size)"""
    import fuzz_parity
    from goldpolish_amd import core
    for name in ("GP_STAGE_A_LOCAL", "GP_STAGE_A_CENSUS", "GP_STAGE_A_CENSUS_MIN"):
        monkeypatch.setenv(name, "0")
    monkeypatch.setenv("GP_WALK_WAVES", "4")
    monkeypatch.setenv("GP_WALK_THREADS", "128")
    with core.Context(0) as ctx:
        for seed in [1086, 20621, 20645, 300029] + list(range(7000, 7009)) + list(range(930000, 930020)):
            assert fuzz_parity.run_case(core, ctx, oracle, seed, "2") is None, seed
