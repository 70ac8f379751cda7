class AttentionBias:   # placeholder type for annotations / isinstance checks in the reference's InputMetadata
    pass
