"""Constants shared by the entry points (counterpart of the reference's common/consts.py:3-33)."""

DS_GEN_SEED = 69                      # dataset split generator seed (consts.py:3)

SUPPORTED_VQ_MODES = ["VectorQuantizer", "GumbelQuantizer", "MultiVectorQuantizer"]

RUN_ID_TIMESTAMP_FORMAT = "%Y_%m_%d_%H_%M_%S"
RUNS_BASE_DIR = "./runs"

# console colours / emoji of the epoch line (cosmetic; same roles as consts.py:13-29)
COLOR_EPOCH = "#BD1376"
COLOR_RUN_ID = COLOR_EPOCH
COLOR_TRAIN, COLOR_VAL, COLOR_TEST = "#2A9CDA", "#6A16A5", "#914418"
COLOR_FROZEN, COLOR_TOT, COLOR_WARNING, COLOR_OFF = "#E71111", "#3C493F", "#b89d0b", "red"
STATS_EMOJI_TRAIN = [":party_popper:", ":rocket:", ":partying_face:", ":fire:"]
STATS_EMOJI_VAL = [":gift:", ":football:", ":dragon:", ":skull:"]
STATS_EMOJI_TEST = [":cowboy_hat_face:", ":crystal_ball:", ":teddy_bear:", ":round_pushpin:"]
