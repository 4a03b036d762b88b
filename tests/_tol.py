"""Tolerances on tile_weights VALUES (the second return value of compute_spatial_entropy, utilities/entropy_utils.py:
131-136, 190-192).

Since round 5 every weights output of the engine comes from ONE producer, the weights pass, whatever formulation produced the
entropy: exact ocml ``acos`` / ``pow`` weights, summed in FP64 — by ``k_weights_gather`` over the plan's exact weight rows (the
users are cut into NW contiguous shares, NW = 4, 2 or 1 by the lattice size and the LDS; each share is summed in column order,
the shares are added in order: deterministic, but not the reference's single sequential sum, hence a relative tolerance and
not bit equality), or, where those rows do not fit the device, by the precise sweep in weights-only mode (strictly column
order).  There is no fixed-point term (round 4 allowed ``users * 2**-33``).  What remains is libm: numpy's and ocml's ``arccos`` differ by
an ulp or so of the distance d, which moves one user's weight ((max - d) / max) ** p by at most
p / max * ulp(d) <= p * 2.2e-16 / max ABSOLUTE (a tile within ~1e-9 rad of the cone's rim has a weight of ~1e-18 that is
all rounding of d in either implementation), and relative 1e-9 everywhere else.
"""
W_RTOL = 1e-9


def w_atol(users: int = 1, power: float = 2.0) -> float:
    """Absolute tolerance on a frame's tile weight sums of ``users`` users: the ulp of arccos, per contributing user.
    (power < 1: d/dr r**p grows like r**(p-1) towards the rim — 1e-12 covers every tile further than 1e-7 rad from it.)"""
    per_user = 4e-16 * float(power) if power >= 1.0 else 1e-12
    return per_user * max(1, int(users))
