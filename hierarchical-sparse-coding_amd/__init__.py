"""MI355X-native convolutional matching-pursuit engine (import it as `hsc_amd`, see ../hsc_amd)."""
