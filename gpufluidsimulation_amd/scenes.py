"""The two synthetic scenes of BASELINE.json's configs, as emitter tuples
(cx, cy, cz, radius, density, temperature, emiter, frames) for BimocqGPUSolver.setSmoke / OracleSolver.set_smoke.
Shared by bench.py, the tests and the golden-hash generators so that all of them run the very same scene."""

# configs 2-4: rising smoke -- one spherical source at step 0, buoyancy along +y (drop 0, rise 1)
SMOKE = (0.5, 0.2, 0.5, 0.1, 1.0, 1.0, 0.0, 1)


def rising_smoke(nz_global, h):
    """the source of SURVEY 8d, centred in z of a grid of nz_global planes of spacing h"""
    return [(SMOKE[0], SMOKE[1], 0.5 * nz_global * h) + SMOKE[3:]]


def leapfrog(nz_global, h):
    """config 5: the reference's vortex-collision emitters (src/bimocq3D/main.cpp:52-78: 10 frames, density 1, the
    velocity ring of emit_smoke_velocity_kernel) placed coaxially and both blowing along +x, so that the rear ring
    threads the front one; no buoyancy (drop = rise = 0)"""
    # The ring axis must not pass through grid nodes: emit_smoke_velocity_kernel normalises the (y, z) offset from the axis
    # (GPU_kernel.cu:750), so a node exactly ON it gets 0/0 -- NaN, for v and w too (NaN * 0) -- and the projection then
    # spreads the NaN 199 cells per step (rounds 1-2 put the axis at (0.5, nz h / 2): on the nodes j = nx/2,
    # k = nz/2 of EVERY grid, so every leapfrog run of those rounds was a NaN run).  The reference nudges its second
    # emitter the same way (main.cpp:64: centre y = 0.201 on a 0.002 grid).
    yc = 0.5 + 0.37 * h
    zc = 0.5 * nz_global * h + 0.29 * h
    return [(0.15, yc, zc, 0.08, 1.0, 0.0, 1.0, 10), (0.35, yc, zc, 0.08, 1.0, 0.0, 1.0, 10)]


def collision(h):
    """the reference binary's own scene (src/bimocq3D/main.cpp:28-80 with the constants BimocqGPUSolver::emitSmoke hard-codes,
    BimocqGPUSolver.cpp:386-389): two spheres of radius 0.015 at x = 0.04 and x = 0.16 of a 0.2 x 0.4 x 0.4 box blowing vortex
    rings at each other (emiter +1 / -1), density 1, temperature 50, 10 frames, no buoyancy.  The first ring's axis, which the
    reference puts at y = z = 0.2 -- node (100, 100) of its 0.002 grid, where the emitter's direction normalisation is 0/0 (SURVEY
    Q14) -- is nudged off the nodes exactly as the reference nudges its second one (y = 0.201)."""
    return [(0.04, 0.2 + 0.37 * h, 0.2 + 0.29 * h, 0.015, 1.0, 50.0, 1.0, 10),
            (0.16, 0.201, 0.2, 0.015, 1.0, 50.0, -1.0, 10)]
