"""Prints the field-product rates plonk_bench_field_mul measures (development aid): the table behind the choice of
multiplier."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa

ctx = pa.Context()
names = {0: "Fr 9 x 29-bit lazy (x_mul)", 1: "Fq 14 x 29-bit lazy (x_mul)", 2: "Fr 8 x 32-bit saturated", 3: "Fq 12 x 32-bit saturated",
         5: "Fq 14 x 29-bit, one Karatsuba level on the product half", 6: "Fq squarings (x_sqr)",
         7: "Fq products in the mix of a mixed addition (7 mul, 2 sqr, 1 mul2 per 10)",
         8: "Fq x_mul with two waves per SIMD (occupancy of msm_bucket_kernel)", 9: "Fq x_mul with one wave per SIMD",
         10: "Fr x_mul with two waves per SIMD", 11: "Fr x_mul with one wave per SIMD"}
mad = ctx.bench_field_mul(4)
print("v_mad_u64_u32: %.2f T lane-op/s" % (mad / 1e12))
for which, name in names.items():
    r = ctx.bench_field_mul(which)
    print("%-75s %7.2f G/s   %6.1f mad-slots per product" % (name, r / 1e9, mad / r))
