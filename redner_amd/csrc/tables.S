/* tables.S -- embeds the two read-only numeric tables into the shared library (.rodata):
 *   Sobol' direction matrices, 1024 dims x 52 u64   (data/sobol_1024x52.u64)
 *   LTC matrices tabM, 128 x 128 x 9 f32            (data/ltc_tabM_128x128x9.f32)
 * Both are produced by tools/extract_tables.py.  Assembled with -I redner_amd/data. */
    .section .rodata
    .balign 64
    .global rdr_sobol_table
rdr_sobol_table:
    .incbin "sobol_1024x52.u64"
    .balign 64
    .global rdr_ltc_table
rdr_ltc_table:
    .incbin "ltc_tabM_128x128x9.f32"
    .section .note.GNU-stack,"",@progbits
