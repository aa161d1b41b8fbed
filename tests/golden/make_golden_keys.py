"""keys of tests/golden/sam2_video_long.npz shared by the generator and the tests (data description only)."""
LONG_MEM_FRAMES = (0, 1, 6, 7, 8, 16, 17)
