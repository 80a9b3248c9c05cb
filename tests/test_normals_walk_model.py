"""The greedy neighbourhood walk of normals.hpp:86-138 over a block's 64 voxel counters, and the loop-free form of it that k_frame_normals
runs (chad_tsdf_amd/csrc/frame_kernels.hpp, estimate_block): both restated here in Python and held against each other on random count
vectors -- neighbourhoods (first point, size) in order, every occupied voxel's neighbourhood, the team lists.  (The device's form itself is
held against the oracle by the GPU tests: test_gpu_golden.py G3, test_gpu_parity.py, tests/soak/fuzz_parity.py.)
Lanes count voxels downwards: lane l is voxel 63 - l, an octant the eight lanes l & ~7 .. l | 7."""
import random

MIN_POINTS, TEAM_POINTS, COOP_POINTS = 8, 32, 256   # kMinPoints (normals.hpp:88), kTeamPoints, kCoopPoints


def _scans(cnt):
    upto, acc = [], 0
    for c in cnt:
        acc += c
        upto.append(acc)
    return upto, [u - c for u, c in zip(upto, cnt)]


def walk_as_written(cnt):
    """the reference's loop: take the highest voxel left; fewer than eight points -> the rest of its octant; still fewer -> the rest of the block"""
    upto, before = _scans(cnt)
    rest = sum(1 << l for l in range(64) if cnt[l])
    groups, mine, narrow, wide, coop = [], [None] * 64, [], [], 0
    while rest:
        l = (rest & -rest).bit_length() - 1
        start, size, end = before[l], cnt[l], l
        if size < MIN_POINTS:
            end = l | 7
            size = upto[end] - start
            if size < MIN_POINTS:
                end = 63
                size = upto[63] - start
        g = len(groups)
        groups.append((start, size))
        if MIN_POINTS <= size < TEAM_POINTS:
            narrow.append(g)
        if TEAM_POINTS <= size < COOP_POINTS:
            wide.append(g)
        coop += size >= COOP_POINTS
        for k in range(l, end + 1):
            mine[k] = g
        rest = 0 if end == 63 else rest & ~((2 << end) - 1)
    return groups, mine, narrow, wide, coop


def walk_without_a_loop(cnt):
    """what every lane of the wave works out for itself from ballots, two cross-lane reads and population counts"""
    upto, before = _scans(cnt)
    popc = lambda m: bin(m).count("1")
    occupied = sum(1 << l for l in range(64) if cnt[l])
    large = sum(1 << l for l in range(64) if cnt[l] >= MIN_POINTS)
    small = occupied & ~large
    tail_lane, tail_size, ending = [64] * 64, [0] * 64, 0
    for l in range(64):
        oct0 = l & ~7
        here = (small >> oct0) & 0xFF
        if here:
            tail_lane[l] = oct0 + (here & -here).bit_length() - 1
        tail_size[l] = upto[oct0 | 7] - before[tail_lane[l] & 63]
        if tail_lane[l] != 64 and tail_size[l] < MIN_POINTS:
            ending |= 1 << l
    last_start = tail_lane[(ending & -ending).bit_length() - 1] if ending else 64
    walked = (1 << 64) - 1 if last_start >= 63 else (2 << last_start) - 1
    alone = [bool((large >> l) & 1) and l < tail_lane[l] for l in range(64)]
    starts = sum(1 << l for l in range(64) if alone[l] or l == tail_lane[l]) & walked
    size = {l: cnt[l] if alone[l] else (upto[63] - before[l] if l == last_start else tail_size[l]) for l in range(64) if (starts >> l) & 1}
    narrow_starts = sum(1 << l for l, s in size.items() if MIN_POINTS <= s < TEAM_POINTS)
    wide_starts = sum(1 << l for l, s in size.items() if TEAM_POINTS <= s < COOP_POINTS)
    groups, narrow, wide = [None] * popc(starts), [None] * popc(narrow_starts), [None] * popc(wide_starts)
    mine = [None] * 64
    for l in range(64):
        below = (1 << l) - 1
        if (starts >> l) & 1:
            g = popc(starts & below)
            groups[g] = (before[l], size[l])
            if (narrow_starts >> l) & 1:
                narrow[popc(narrow_starts & below)] = g
            if (wide_starts >> l) & 1:
                wide[popc(wide_starts & below)] = g
        cover = last_start if last_start != 64 and l >= last_start else tail_lane[l] if tail_lane[l] != 64 and l >= tail_lane[l] else l
        mine[l] = popc(starts & ((1 << cover) - 1))
    return groups, mine, narrow, wide, sum(s >= COOP_POINTS for s in size.values())


def test_the_walk_without_its_loop_equals_the_walk_as_written():
    rng = random.Random(5)
    for trial in range(12000):
        mode = rng.random()
        cnt = []
        for _ in range(64):
            r = rng.random()
            if mode < 0.3:
                cnt.append(0 if r < 0.5 else rng.randint(1, 12))
            elif mode < 0.6:
                cnt.append(0 if r < 0.8 else rng.randint(1, 40))
            elif mode < 0.8:
                cnt.append(0 if r < 0.2 else rng.randint(1, 9))
            else:
                cnt.append(0 if r < 0.9 else rng.choice([1, 2, 7, 8, 31, 32, 255, 256, 400]))
        a, b = walk_as_written(cnt), walk_without_a_loop(cnt)
        assert a[0] == b[0] and a[2:] == b[2:], cnt
        for l in range(64):   # every occupied voxel's neighbourhood, and every empty voxel's that lies inside a tail
            if cnt[l] or a[1][l] is not None:
                assert a[1][l] == b[1][l], (cnt, l)


def test_walk_corner_cases():
    for cnt in ([0] * 64, [1] + [0] * 63, [0] * 63 + [1], [8] * 64, [7] * 64, [1] * 64, [0] * 56 + [7] + [0] * 6 + [1], [300] + [0] * 62 + [3]):
        assert walk_as_written(cnt)[0] == walk_without_a_loop(cnt)[0] and walk_as_written(cnt)[2:] == walk_without_a_loop(cnt)[2:], cnt
