from .step import STEP
from .tsformer import TSFormer
from .graphwavenet import GraphWaveNet
from .discrete_graph_learning import DiscreteGraphLearning

__all__ = ["STEP", "TSFormer", "GraphWaveNet", "DiscreteGraphLearning"]
