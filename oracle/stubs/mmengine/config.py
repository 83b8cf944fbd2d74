class Config(dict):
    @staticmethod
    def fromstring(s, file_format='.py'):
        raise NotImplementedError('stub')
