"""Mirror of the reference's `models` package: only the plug-in seam (YAML `target: models.codec_sq_fixbpp.Codec`)."""
