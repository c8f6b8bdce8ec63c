from efficientat_amd.dymn import (ContextGen, CoordAtt, DY_Block, DynamicConv, DynamicInvertedResidualConfig,  # noqa: F401
                                  DyReLUB)
