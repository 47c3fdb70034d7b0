def getDataPath():
    return ""
