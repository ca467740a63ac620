from .pangu import PanguModel

# The reference registers pangu, fourcastnet, fourcastnet_v2, dlwp, graphcast, fuxi, fengwu
# (/root/reference/skyrim/core/models/__init__.py:9-17).  This build ships the hot path of one of
# them; the others are later rows of SURVEY.md 8 and are absent rather than stubbed.
MODELS = {
    "pangu": PanguModel,
}
