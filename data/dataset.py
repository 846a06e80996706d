"""`data.dataset` names the configs reference (["data.dataset", "PairedDataset"], config/noise.json:88)."""
from mdie_amd.host import ImageFolder as UnpairedDataset  # noqa: F401
from mdie_amd.host import ImageFolderPairs as PairedDataset  # noqa: F401
