/* The constant data of RFC 7932 (static dictionary, transforms, context lookup: tools/gen_brotli_data.py)
 * linked into the library; otz_kernels.hip uploads it to a device the first time a batch needs it. */
	.section .rodata
	.balign 16
	.global otz_brotli_blob
	.global otz_brotli_blob_end
otz_brotli_blob:
	.incbin "brotli_rfc7932.bin"
otz_brotli_blob_end:
	.section .note.GNU-stack,"",@progbits
