#!/usr/bin/env python3
"""Sample counts of Camera::multisample (rust/src/camera.rs:273-310) for pixels of CONSTANT colour, derived from the
reference source by executing its statements one by one in IEEE f64 (Python floats), no renderer involved.

For a pixel whose every sample is c the loop's state has a closed form - sum = n c, sos = c + (n - 1) c^2 because
camera.rs:277 seeds the sum of squares with the first sample UN-squared - and the loop stops at the first n >=
min_samples with mean_rgb(c - c^2) / n^2 <= SE^2, i.e. n ~ sqrt(mean(c - c^2)) / SE: 50 samples for mid-grey at
SE = 1e-2, where a sum of squares seeded with c * c would stop at min_samples.  The statements below are the
reference's, in its order and association (Color / f64 is Color::scale(1.0 / n), color.rs:145-149), so that cases on an
exact boundary (c = 0.5: n = 50 exactly) come out as the reference's own rounding decides.

Writes tests/golden/multisample_counts.json.  Run by hand; the output is committed.
"""
import json
import os


def color_variance_of_mean(n, s, sos):  # camera.rs:305-310
    inv = 1.0 / n
    mean = [x * inv for x in s]                       # let mean_rgb = sum / n;
    q = [x * inv for x in sos]
    var = [q[i] - mean[i] * mean[i] for i in range(3)]  # let variance_rgb = sos / n - mean_rgb * mean_rgb;
    variance = (var[0] + var[1] + var[2]) / 3.0
    return variance / n


def multisample_constant(c, se, min_samples):  # camera.rs:273-302 with world.trace(...) == c for every ray
    sos = list(c)   # let mut color_sum_of_squares = c;      (line 277: NOT c * c)
    s = list(c)     # let mut color_sum = c;
    n = 1.0
    while n < float(min_samples):
        s = [s[i] + c[i] for i in range(3)]
        sos = [sos[i] + c[i] * c[i] for i in range(3)]
        n += 1.0
    while color_variance_of_mean(n, s, sos) > se * se:
        s = [s[i] + c[i] for i in range(3)]
        sos = [sos[i] + c[i] * c[i] for i in range(3)]
        n += 1.0
    return int(n), [x * (1.0 / n) for x in s]


CASES = [
    # (name, colour, allowed_standard_error, min_samples, on an exact boundary of the rule in real arithmetic)
    ("mid grey as the examples ship (SE 1e-2, min 5)", [0.5, 0.5, 0.5], 1e-2, 5, True),
    ("0.2 grey", [0.2, 0.2, 0.2], 1e-2, 5, True),
    ("0.3 grey", [0.3, 0.3, 0.3], 1e-2, 5, False),
    ("0.9 grey", [0.9, 0.9, 0.9], 1e-2, 5, True),
    ("chapter-11 sky colour 0.8 0.8 1.0", [0.8, 0.8, 1.0], 1e-2, 5, False),
    ("checker colour 0.75 0.25 0.5", [0.75, 0.25, 0.5], 1e-2, 5, False),
    ("over-bright 2.0: c - c^2 < 0, stops at min_samples", [2.0, 2.0, 2.0], 1e-2, 5, False),
    ("black: stops at min_samples", [0.0, 0.0, 0.0], 1e-2, 5, False),
    ("white: stops at min_samples", [1.0, 1.0, 1.0], 1e-2, 5, False),
    ("Camera::new defaults (SE 0.1, min 5), 0.3 grey", [0.3, 0.3, 0.3], 0.1, 5, False),
    ("render_interactive (camera.rs:182: min 1, SE 0.1), 0.3 grey", [0.3, 0.3, 0.3], 0.1, 1, False),
    ("render_interactive, mid grey", [0.5, 0.5, 0.5], 0.1, 1, True),
    ("deterministic set-up of camera.rs:378-389 (min 1, SE 1.0)", [0.5, 0.5, 0.5], 1.0, 1, False),
    ("depth-of-field example (SE 5e-3), 0.3 grey", [0.3, 0.3, 0.3], 5e-3, 5, False),
    ("min 10 as chapter-15 ships, 0.03 grey", [0.03, 0.03, 0.03], 1e-2, 10, False),
]

if __name__ == "__main__":
    out = []
    for name, c, se, mn, boundary in CASES:
        n, mean = multisample_constant(c, se, mn)
        out.append({"name": name, "color": c, "allowed_standard_error": se, "min_samples": mn, "samples": n, "mean": mean,
                    "boundary": boundary})
        print("%-70s n = %d" % (name, n))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multisample_counts.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
