"""Run-models with the reference's module protocol: `version`, `init(base_model) -> (H, W)`, `run(parsed_layout, seed, ...)`
(generation/lvd.py:12,19-53,85-196; consumed by generate.py:125-165,331-338)."""
