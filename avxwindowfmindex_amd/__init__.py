"""MI355X-native FM-index k-mer search behind the AwFmIndex.h C API."""
